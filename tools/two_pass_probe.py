"""tools/two_pass_probe.py -- what two seeding passes in flight on one GPU would give: two engines (each with its own copy of the index, so
a smaller genome) seed device-resident batches from two host threads; aggregate rate against one engine alone.
usage: two_pass_probe.py [genome_mbp] [reads] [engines]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import compseed_amd as ca, synth
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
NE = int(sys.argv[3]) if len(sys.argv) > 3 else 2
G = synth.make_genome(int(mbp * 1e6), seed=20261003, device="cuda")
ix = ca.Index.build(G.cpu().numpy(), 0)
bases, off = synth.make_reads(G, n, 150, seed=777, p_sub=0.005, sort=True)
del G; torch.cuda.empty_cache()
engs = [ca.Engine(ix, 0, sa64=1) for _ in range(NE)]
par = ca.Params()
K = 6


def run(e, k, nn):
    for _ in range(k):
        e.seed_batch_device(bases.data_ptr(), off.data_ptr(), nn, nn * 150, par)


for nn in (n, n // 2, n // 4):
    for e in engs:
        run(e, 2, nn)
    torch.cuda.synchronize()
    t = time.perf_counter(); run(engs[0], K, nn); dt1 = (time.perf_counter() - t) / K
    th = [threading.Thread(target=run, args=(e, K, nn)) for e in engs]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    dt2 = (time.perf_counter() - t) / (NE * K)
    print("%d reads per pass: one engine %.1f ms per pass (%.1f M reads/s); %d engines at once %.1f ms per pass (%.1f M reads/s)" % (nn, dt1 * 1e3, nn / dt1 / 1e6, NE, dt2 * 1e3, nn / dt2 / 1e6), flush=True)
