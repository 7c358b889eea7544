"""tools/integration_time.py -- wall time of the unpatched reference (integration/_build/CompSeed.ref) and of the patched one that takes its
seeds, chains and alignment regions from the library (CompSeed.gpu), -t 16, SAM compared.  Reads: N of 150 bp sampled from the golden reference
(220 kbp with tandem arrays: 68 alignment regions per read, the quadratic passes' worst case) or, with --synth-mbp M, from a synthetic genome of
M Mbp (tools/synth.py, the bench's default profile) indexed by cs_index_build_fasta (the reference reads the same five files).
usage: integration_time.py [reads] [--synth-mbp M]"""
import argparse, gzip, hashlib, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import _data

ap = argparse.ArgumentParser()
ap.add_argument("reads", nargs="?", type=int, default=300000)
ap.add_argument("--synth-mbp", type=float, default=0.0)
a = ap.parse_args()
n = a.reads
L = 150
idx_dir = None
if a.synth_mbp > 0:
    import torch, synth
    import compseed_amd as ca
    G = synth.make_genome(int(a.synth_mbp * 1e6), seed=20261003, device="cuda")
    rd, ro = synth.make_reads(G, n, L, seed=99, p_sub=0.005, p_indel=0.001, sort=True)
    rb = np.frombuffer(b"ACGT", dtype=np.uint8)[rd.cpu().numpy()] if rd.max().item() < 4 else rd.cpu().numpy()
    off = ro.cpu().numpy().astype(np.int64)
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[G.cpu().numpy()]
    del G, rd, ro; torch.cuda.empty_cache()
    idx_dir = tempfile.mkdtemp(prefix="csint_")
    fa = os.path.join(idx_dir, "g.fa")
    with open(fa, "wb") as f:
        nctg = 8; per = (g.size + nctg - 1) // nctg
        for k in range(nctg):
            f.write(b">chr%d\n" % (k + 1)); f.write(g[k * per:(k + 1) * per].tobytes()); f.write(b"\n")
    del g
    PREFIX = os.path.join(idx_dir, "g")
    ca.build_index_from_fasta(fa, PREFIX, 0)
    os.remove(fa)
    lens = np.diff(off)
    if (lens == L).all():
        reads = np.empty((n, L + 1), dtype=np.uint8); reads[:, :L] = rb.reshape(n, L); reads[:, L] = 10
        blob = reads.tobytes()
    else:
        blob = b"".join(rb[off[i]:off[i + 1]].tobytes() + b"\n" for i in range(n))
else:
    PREFIX = _data.PREFIX
    fa = gzip.open(os.path.join(_data.GOLD, "ref.fa.gz")).read().decode().split(">")[1:]
    contigs = [np.frombuffer("".join(c.split("\n")[1:]).upper().replace("N", "A").encode(), dtype=np.uint8) for c in fa]
    rng = np.random.default_rng(11)
    ci = rng.integers(0, len(contigs), n)
    reads = np.empty((n, L + 1), dtype=np.uint8); reads[:, L] = 10
    for k, c in enumerate(contigs):
        sel = np.nonzero(ci == k)[0]
        p = rng.integers(0, c.size - L - 8, sel.size)
        reads[sel, :L] = c[p[:, None] + np.arange(L)[None, :]]
    mut = rng.random((n, L)) < 0.01
    sub = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(mut.sum()))]
    body = reads[:, :L]; body[mut] = sub
    blob = reads.tobytes()
out = {"reads": n, "threads": 16, "reference": ("synthetic %g Mbp" % a.synth_mbp) if a.synth_mbp > 0 else "tests/golden/g1 (220 kbp, tandem arrays)"}
with tempfile.TemporaryDirectory() as td:
    fn = os.path.join(td, "reads.txt"); open(fn, "wb").write(blob)
    for name in ("CompSeed.ref", "CompSeed.gpu"):
        exe = os.path.join(ROOT, "integration", "_build", name)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-t", "16", PREFIX, fn], capture_output=True, cwd=td, timeout=1500)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-1500:]
        err = r.stderr.decode(errors="replace")
        out[name] = {"seconds": dt, "reads_per_s": n / dt, "sam_md5": hashlib.md5(r.stdout).hexdigest(), "sam_bytes": len(r.stdout),
                     "stderr_tail": [l for l in err.splitlines() if l.startswith(("GPU", "Wall", "BWT", "SA "))][-6:]}
out["same_sam"] = out["CompSeed.ref"]["sam_md5"] == out["CompSeed.gpu"]["sam_md5"]
if idx_dir: shutil.rmtree(idx_dir, ignore_errors=True)
print(json.dumps(out))
