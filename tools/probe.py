"""tools/probe.py -- access-shape ceiling: dependent random 64-B block reads on a synthetic index (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import compseed_amd as ca
import synth
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 1000
G = synth.make_genome(int(mbp * 1e6), seed=20261003, device="cuda")
ix = ca.Index.build(G.cpu().numpy(), 0)
del G; torch.cuda.empty_cache()
eng = ca.Engine(ix, 0)
print("index: %.2f GB of Occ blocks" % (ix.view.bwt_size * 4 / 1e9))
for w in (1, 2, 4, 8):
    eng.probe_random_lines(w, 200)
    r = eng.probe_random_lines(w, 2000)
    lanes = 256 * 4 * w * 64
    print("waves/SIMD %d: %.2f G lines/s = %.2f TB/s ; per-lane step latency %.2f us" % (w, r / 1e9, r * 64 / 1e12, lanes / r * 1e6))

# workload shape from the oracle (CPU) on a sample of the bench reads
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, _oracle
Gd = synth.make_genome(int(mbp * 1e6), seed=20261003, device="cuda")
bases, off = synth.make_reads(Gd, 200000, 150, seed=777, p_sub=0.005, sort=True)
bw, sa = ix.arrays()
o = _oracle.OracleIndex.from_arrays(ix.view.primary, [ix.view.L2[i] for i in range(1, 5)], bw, sa, 32)
st = o.seed_batch(bases.cpu().numpy(), off.cpu().numpy().astype(np.uint64), mode=0, want_sal=True, threads=16)["stats"]
n = 200000
print({k: (round(v / n, 2) if isinstance(v, int) else v) for k, v in st.items()})
