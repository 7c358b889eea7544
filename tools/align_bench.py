"""tools/align_bench.py -- the stages behind seeding at a size where their speed shows: N reads of 150 bases sampled from the golden
reference (tests/golden/g1, 220 kbp: every read has a true locus, many have repeats) with substitutions and short indels, through
GPU seeding -> cs_chain_batch -> cs_chain_filter -> cs_extend_chains -> cs_dedup_regions; wall time and reads/s per stage.
usage: align_bench.py [reads] [--synth-mbp M]"""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compseed_amd as ca
import _data

import argparse, tempfile, shutil


def run(n=200000, synth_mbp=0.0):
    L = 150
    tmpdir = None
    if synth_mbp > 0:
        import torch
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import synth
        G = synth.make_genome(int(synth_mbp * 1e6), seed=20261003, device="cuda")
        rd, ro = synth.make_reads(G, n, L, seed=99, p_sub=0.005, p_indel=0.001, sort=True)
        bases = rd.cpu().numpy(); off = ro.cpu().numpy().astype(np.uint64)
        g = np.frombuffer(b"ACGT", dtype=np.uint8)[G.cpu().numpy()]
        del G, rd, ro; torch.cuda.empty_cache()
        tmpdir = tempfile.mkdtemp(prefix="csaln_")
        fa = os.path.join(tmpdir, "g.fa")
        with open(fa, "wb") as f:
            nctg = 8; per = (g.size + nctg - 1) // nctg
            for k in range(nctg):
                f.write(b">chr%d\n" % (k + 1)); f.write(g[k * per:(k + 1) * per].tobytes()); f.write(b"\n")
        del g
        PREFIX = os.path.join(tmpdir, "g")
        t0 = time.perf_counter(); ca.build_index_from_fasta(fa, PREFIX, 0); t_build = time.perf_counter() - t0
        os.remove(fa)
    else:
        PREFIX = _data.PREFIX
        fa = gzip.open(os.path.join(_data.GOLD, "ref.fa.gz")).read().decode().split(">")[1:]
        contigs = [np.frombuffer("".join(c.split("\n")[1:]).upper().replace("N", "A").encode(), dtype=np.uint8) for c in fa]
        rng = np.random.default_rng(3)
        ci = rng.integers(0, len(contigs), n)
        reads = np.empty((n, L), dtype=np.uint8)
        for k, c in enumerate(contigs):
            sel = np.nonzero(ci == k)[0]
            p = rng.integers(0, c.size - L - 8, sel.size)
            reads[sel] = c[p[:, None] + np.arange(L)[None, :]]
        mut = rng.random((n, L)) < 0.01
        reads[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(mut.sum()))]
        # a 1-2-base deletion in one read of four (the tail shifts left, the end is refilled with random bases)
        for r in np.nonzero(rng.random(n) < 0.25)[0][:50000]:
            at = int(rng.integers(40, 110)); k = int(rng.integers(1, 3))
            reads[r, at:L - k] = reads[r, at + k:]
            reads[r, L - k:] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, k)]
        rc = rng.random(n) < 0.5
        comp = np.zeros(256, np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
        reads[rc] = comp[reads[rc]][:, ::-1]
        bases = np.ascontiguousarray(reads.reshape(-1)); off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    ix = ca.Index.load(PREFIX); eng = ca.Engine(ix, 0); ch = ca.Chainer(PREFIX); al = ca.Aligner(PREFIX, 0)
    out = {"reads": n, "reference": ("synthetic %g Mbp" % synth_mbp) if synth_mbp > 0 else "tests/golden/g1 (220 kbp, tandem arrays)"}
    for rep in range(2):   # the second round is the measured one (buffers sized)
        t0 = time.perf_counter(); res = eng.seed_batch(bases, off, ca.Params(), copy=False); t1 = time.perf_counter()
        c = ch.chain(res.mem_off, res.mems, res.seed_off, res.seeds, off, ca.ChainParams(), threads=16, copy=False); t2 = time.perf_counter()
        f = ch.filter(c["chain_off"], c["chains"], c["cseed_off"], c["cseeds"], bases, off, threads=16, copy=False); t3 = time.perf_counter()
        st0 = al.stats()
        g = al.extend_chains(f["chain_off"], f["chains"], f["cseed_off"], f["cseeds"], bases, off, cseed_score=f["cseed_score"], copy=False); t4 = time.perf_counter()
        st1 = al.stats()
        d = al.dedup_regions(g["reg_off"], g["regs"], bases, off, copy=False); t5 = time.perf_counter()
    for name, a, b in (("seed (host call)", t0, t1), ("chain", t1, t2), ("chain_filter", t2, t3), ("extend_chains", t3, t4), ("dedup_regions", t4, t5)):
        out[name] = {"ms": 1e3 * (b - a), "reads_per_s": n / (b - a)}
    out["counts"] = {"seeds": int(res.n_seeds), "chains": int(c["chains"].size), "chains_after_filter": int(f["chains"].size), "regions": int(g["regs"].size),
                     "regions_after_dedup": int(d["regs"].size), "extensions": int(st1["pairs"] - st0["pairs"]), "ext_launches": int(st1["launches"] - st0["launches"])}
    per_read = np.diff(np.asarray(g["reg_off"]).astype(np.int64))
    out["regions_per_read"] = {"mean": float(per_read.mean()), "p50": float(np.percentile(per_read, 50)), "p99": float(np.percentile(per_read, 99)), "p99.9": float(np.percentile(per_read, 99.9)),
                               "max": int(per_read.max()), "reads_over_64": int((per_read > 64).sum()), "reads_over_1024": int((per_read > 1024).sum()),
                               "sum_sq_over_64": float(((per_read.astype(np.float64) ** 2) / 64).sum())}
    kms = st1["ext_kernel_ms"] - st0["ext_kernel_ms"]
    out["extension_kernels"] = {"ms": kms, "pairs_per_s": (st1["pairs"] - st0["pairs"]) / (kms * 1e-3) if kms > 0 else None, "dp_cells_per_pair": (st1["ext_cells"] - st0["ext_cells"]) / max(1, st1["pairs"] - st0["pairs"]),
                                "note": "the extension kernels alone (HIP events) on this run's own pairs: extensions of real chains die at different rows, unlike the uniform pairs of tools/extend_bench.py"}
    t_all = sum(out[k]["ms"] for k in ("chain", "chain_filter", "extend_chains", "dedup_regions"))
    out["behind_seeding_reads_per_s"] = n / (t_all * 1e-3)
    if tmpdir:
        shutil.rmtree(tmpdir, ignore_errors=True)
    for x in (al, ch, eng, ix):
        x.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("reads", nargs="?", type=int, default=200000)
    ap.add_argument("--synth-mbp", type=float, default=0.0, help="instead of the golden reference: a synthetic genome of this size (tools/synth.py, the bench's default profile: random + a 10 %% "
                    "repeat family + duplications), written as FASTA (8 contigs), indexed on the GPU (cs_index_build_fasta); reads with 0.5 %% substitutions, one read in seven with a single-base indel")
    a = ap.parse_args()
    print(json.dumps(run(a.reads, a.synth_mbp)))
