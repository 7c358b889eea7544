"""tools/host_stage_scaling.py -- the threaded host passes behind seeding (cs_chain_batch, cs_chain_filter, cs_dedup_regions) on 1 .. 64 threads:
1 M reads of 150 bp on a 200 Mbp synthetic genome (the workload of bench.py's `extension.stage`), wall time per pass.  Run on the GPU box (seeding
and the extension in between run on the GPU)."""
import sys, os, time, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tools"); sys.path.insert(0, R + "/tests")
import numpy as np, torch
import compseed_amd as ca, synth, tempfile
n=1000000; L=150
G = synth.make_genome(int(200e6), seed=20261003, device="cuda")
rd, ro = synth.make_reads(G, n, L, seed=99, p_sub=0.005, p_indel=0.001, sort=True)
bases = rd.cpu().numpy(); off = ro.cpu().numpy().astype(np.uint64)
g = np.frombuffer(b"ACGT", dtype=np.uint8)[G.cpu().numpy()]
del G, rd, ro; torch.cuda.empty_cache()
tmpdir = tempfile.mkdtemp(prefix="csaln_"); fa = os.path.join(tmpdir, "g.fa")
with open(fa, "wb") as f:
    nctg = 8; per = (g.size + nctg - 1) // nctg
    for k in range(nctg):
        f.write(b">chr%d\n" % (k + 1)); f.write(g[k * per:(k + 1) * per].tobytes()); f.write(b"\n")
PREFIX = os.path.join(tmpdir, "g"); ca.build_index_from_fasta(fa, PREFIX, 0)
ix = ca.Index.load(PREFIX); eng = ca.Engine(ix, 0); ch = ca.Chainer(PREFIX); al = ca.Aligner(PREFIX, 0)
res = eng.seed_batch(bases, off, ca.Params(), copy=False)
for T in (1, 4, 8, 16, 32, 64):
    ts = []
    for rep in range(2):
        t0 = time.perf_counter(); c = ch.chain(res.mem_off, res.mems, res.seed_off, res.seeds, off, ca.ChainParams(), threads=T, copy=False); t1 = time.perf_counter()
        f = ch.filter(c["chain_off"], c["chains"], c["cseed_off"], c["cseeds"], bases, off, threads=T, copy=False); t2 = time.perf_counter()
        ts = [t1 - t0, t2 - t1]
    print("threads %3d: chain %7.1f ms  filter %7.1f ms" % (T, ts[0] * 1e3, ts[1] * 1e3), flush=True)
gg = al.extend_chains(f["chain_off"], f["chains"], f["cseed_off"], f["cseeds"], bases, off, cseed_score=f["cseed_score"], copy=False)
for T in (1, 4, 8, 16, 32, 64):
    al2 = ca.Aligner(PREFIX, -1, ca.AlnParams(threads=T))
    for rep in range(2):
        t0 = time.perf_counter(); d = al2.dedup_regions(gg["reg_off"], gg["regs"], bases, off, copy=False); t1 = time.perf_counter()
    print("threads %3d: dedup %7.1f ms" % (T, (t1 - t0) * 1e3), flush=True)
