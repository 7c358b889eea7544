"""tools/integration_time.py -- wall time of the unpatched reference (integration/_build/CompSeed.ref) and of the patched one that takes its
seeds, chains and alignment regions from the library (CompSeed.gpu) on N reads sampled from the golden reference, -t 16; SAM compared.
usage: integration_time.py [reads]"""
import gzip, hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
fa = gzip.open(os.path.join(_data.GOLD, "ref.fa.gz")).read().decode().split(">")[1:]
contigs = [np.frombuffer("".join(c.split("\n")[1:]).upper().replace("N", "A").encode(), dtype=np.uint8) for c in fa]
rng = np.random.default_rng(11)
L = 150
ci = rng.integers(0, len(contigs), n)
reads = np.empty((n, L + 1), dtype=np.uint8); reads[:, L] = 10
for k, c in enumerate(contigs):
    sel = np.nonzero(ci == k)[0]
    p = rng.integers(0, c.size - L - 8, sel.size)
    reads[sel, :L] = c[p[:, None] + np.arange(L)[None, :]]
mut = rng.random((n, L)) < 0.01
sub = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(mut.sum()))]
body = reads[:, :L]; body[mut] = sub
out = {"reads": n, "threads": 16}
with tempfile.TemporaryDirectory() as td:
    fn = os.path.join(td, "reads.txt"); open(fn, "wb").write(reads.tobytes())
    for name in ("CompSeed.ref", "CompSeed.gpu"):
        exe = os.path.join(ROOT, "integration", "_build", name)
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-t", "16", _data.PREFIX, fn], capture_output=True, cwd=td, timeout=1500)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-1500:]
        err = r.stderr.decode(errors="replace")
        out[name] = {"seconds": dt, "reads_per_s": n / dt, "sam_md5": hashlib.md5(r.stdout).hexdigest(), "sam_bytes": len(r.stdout),
                     "stderr_tail": [l for l in err.splitlines() if l.startswith(("GPU", "Wall", "BWT", "SA "))][-6:]}
out["same_sam"] = out["CompSeed.ref"]["sam_md5"] == out["CompSeed.gpu"]["sam_md5"]
print(json.dumps(out))
