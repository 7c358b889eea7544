"""tools/hostpipe_probe.py -- timeline of the host pipeline at bench scale (engine log at verbose = 2): one blocking packed call, one
expanded call, and a stream of batches with three in flight.  usage: hostpipe_probe.py [genome_mbp] [reads] [key=value engine options ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import compseed_amd as ca, synth
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
opts = {k: int(v) for k, v in (a.split("=") for a in sys.argv[3:])}
G = synth.make_genome(int(mbp * 1e6), seed=20261003, device="cuda")
ix = ca.Index.build(G.cpu().numpy(), 0)
bases, off = synth.make_reads(G, n, 150, seed=777, p_sub=0.005, sort=True)
del G; torch.cuda.empty_cache()
pin = ca.pinned_array(bases.numel()); pin[:] = bases.cpu().numpy(); ho = off.cpu().numpy().astype(np.uint64)
eng = ca.Engine(ix, 0, verbose=0, **opts)
for _ in range(3):
    eng.seed_batch_packed(pin, ho)                                       # sizes the buffers (three result slots)
eng.close()
eng = ca.Engine(ix, 0, verbose=2, **opts)
for _ in range(3):
    eng.seed_batch_packed(pin, ho)
print("---- one blocking packed call", flush=True); sys.stderr.flush()
t = time.perf_counter(); eng.seed_batch_packed(pin, ho); dt = time.perf_counter() - t
sys.stderr.flush(); print("packed: %.1f ms -> %.1f M reads/s" % (dt * 1e3, n / dt / 1e6), flush=True)
eng.seed_batch(pin, ho, copy=False)
print("---- one blocking expanded call", flush=True)
t = time.perf_counter(); eng.seed_batch(pin, ho, copy=False); dt = time.perf_counter() - t
sys.stderr.flush(); print("expanded: %.1f ms -> %.1f M reads/s" % (dt * 1e3, n / dt / 1e6), flush=True)
print("---- stream, four in flight", flush=True)
depth, total = 4, 24
for _ in range(depth):
    eng.submit(pin, ho)
tt = []
for i in range(total):
    eng.collect_packed(); tt.append(time.perf_counter())
    if i + depth < total:
        eng.submit(pin, ho)
dt = (tt[20] - tt[8]) / 12
sys.stderr.flush(); print("stream of batches, four in flight: %.1f ms per batch -> %.1f M reads/s" % (dt * 1e3, n / dt / 1e6), flush=True)
eng.close()
