"""tools/align_bench.py -- the stages behind seeding at a size where their speed shows: N reads of 150 bases sampled from the golden
reference (tests/golden/g1, 220 kbp: every read has a true locus, many have repeats) with substitutions and short indels, through
GPU seeding -> cs_chain_batch -> cs_chain_filter -> cs_extend_chains -> cs_dedup_regions; wall time and reads/s per stage.
usage: align_bench.py [reads]"""
import gzip, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import compseed_amd as ca
import _data

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
fa = gzip.open(os.path.join(_data.GOLD, "ref.fa.gz")).read().decode().split(">")[1:]
contigs = [np.frombuffer("".join(c.split("\n")[1:]).upper().replace("N", "A").encode(), dtype=np.uint8) for c in fa]
rng = np.random.default_rng(3)
L = 150
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
ix = ca.Index.load(_data.PREFIX); eng = ca.Engine(ix, 0); ch = ca.Chainer(_data.PREFIX); al = ca.Aligner(_data.PREFIX, 0)
out = {"reads": n}
for rep in range(2):   # the second round is the measured one (buffers sized)
    t0 = time.perf_counter(); res = eng.seed_batch(bases, off, ca.Params()); t1 = time.perf_counter()
    c = ch.chain(res.mem_off, res.mems, res.seed_off, res.seeds, off, ca.ChainParams(), threads=16); t2 = time.perf_counter()
    f = ch.filter(c["chain_off"], c["chains"], c["cseed_off"], c["cseeds"], bases, off, threads=16); t3 = time.perf_counter()
    st0 = al.stats()
    g = al.extend_chains(f["chain_off"], f["chains"], f["cseed_off"], f["cseeds"], bases, off, cseed_score=f["cseed_score"]); t4 = time.perf_counter()
    st1 = al.stats()
    d = al.dedup_regions(g["reg_off"], g["regs"], bases, off); t5 = time.perf_counter()
for name, a, b in (("seed (host call)", t0, t1), ("chain", t1, t2), ("chain_filter", t2, t3), ("extend_chains", t3, t4), ("dedup_regions", t4, t5)):
    out[name] = {"ms": 1e3 * (b - a), "reads_per_s": n / (b - a)}
out["counts"] = {"seeds": int(res.n_seeds), "chains": int(c["chains"].size), "chains_after_filter": int(f["chains"].size), "regions": int(g["regs"].size),
                 "regions_after_dedup": int(d["regs"].size), "extensions": int(st1["pairs"] - st0["pairs"]), "ext_launches": int(st1["launches"] - st0["launches"])}
print(json.dumps(out))
