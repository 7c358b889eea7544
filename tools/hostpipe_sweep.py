"""tools/hostpipe_sweep.py -- blocking packed call and stream of batches at bench scale for several engine option sets.
usage: hostpipe_sweep.py "k=v,k=v" "k=v" ...   (each argument one option set; "" = defaults)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import compseed_amd as ca, synth
mbp, n = 3100, 10_000_000
G = synth.make_genome(int(mbp * 1e6), seed=20261003, device="cuda")
ix = ca.Index.build(G.cpu().numpy(), 0)
bases, off = synth.make_reads(G, n, 150, seed=777, p_sub=0.005, sort=True)
del G; torch.cuda.empty_cache()
pin = ca.pinned_array(bases.numel()); pin[:] = bases.cpu().numpy(); ho = off.cpu().numpy().astype(np.uint64)
for spec in sys.argv[1:] or [""]:
    opts = {k: int(v) for k, v in (a.split("=") for a in spec.split(",") if a)}
    eng = ca.Engine(ix, 0, **opts)
    for _ in range(3):
        eng.seed_batch_packed(pin, ho)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); eng.seed_batch_packed(pin, ho); ts.append((time.perf_counter() - t) * 1e3)
    depth = 4
    for _ in range(depth):
        eng.submit(pin, ho)
    tt = []
    NB = 64                              # batches in the stream; timed: the 40 in the middle
    for i in range(NB):
        eng.collect_packed(); tt.append(time.perf_counter())
        if i + depth < NB:
            eng.submit(pin, ho)
    per = (tt[52] - tt[12]) / 40 * 1e3
    gaps = np.diff(np.array(tt[12:53])) * 1e3
    print("%-50s blocking packed: min %.1f median %.1f ms (%.1f M reads/s) | stream: %.1f ms per batch (%.1f M reads/s; between collects: median %.1f, 10th-90th percentile %.1f-%.1f ms)" %
          (spec or "defaults", min(ts), sorted(ts)[2], n / sorted(ts)[2] / 1e3, per, n / per / 1e3, np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90)), flush=True)
    eng.close()
