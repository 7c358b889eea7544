"""tools/hostpipe_probe.py -- timeline of the host pipeline (cs_engine_seed_batch_packed) at bench scale, for several sub-batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import compseed_amd as ca, synth
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
G = synth.make_genome(int(mbp * 1e6), seed=20261003, device="cuda")
ix = ca.Index.build(G.cpu().numpy(), 0)
bases, off = synth.make_reads(G, n, 150, seed=777, p_sub=0.005, sort=True)
del G; torch.cuda.empty_cache()
pin = ca.pinned_array(bases.numel()); pin[:] = bases.cpu().numpy(); ho = off.cpu().numpy().astype(np.uint64)
for pr in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "5000000", "2500000", "1250000"])]:
    eng = ca.Engine(ix, 0, pipeline_reads=pr, verbose=0)
    eng.seed_batch_packed(pin, ho)
    eng.close()
    eng = ca.Engine(ix, 0, pipeline_reads=pr, verbose=1)
    eng.seed_batch_packed(pin, ho); eng.seed_batch_packed(pin, ho)      # sizes the buffers (two result slots)
    print("---- pipeline_reads", pr, flush=True); sys.stderr.flush()
    t = time.perf_counter(); eng.seed_batch_packed(pin, ho); dt = time.perf_counter() - t
    print("packed: %.1f ms -> %.1f M reads/s" % (dt * 1e3, n / dt / 1e6), flush=True)
    eng.seed_batch(pin, ho, copy=False)
    t = time.perf_counter(); eng.seed_batch(pin, ho, copy=False); dt = time.perf_counter() - t
    print("expanded: %.1f ms -> %.1f M reads/s" % (dt * 1e3, n / dt / 1e6), flush=True)
    eng.submit(pin, ho); eng.submit(pin, ho); eng.collect_packed(); eng.submit(pin, ho)
    t = time.perf_counter()
    for i in range(4):
        eng.collect_packed()
        if i < 3:
            eng.submit(pin, ho)
    dt = (time.perf_counter() - t) / 4
    eng.collect_packed()
    print("stream of batches, two in flight: %.1f ms per batch -> %.1f M reads/s" % (dt * 1e3, n / dt / 1e6), flush=True)
    eng.close()
