#!/usr/bin/env python3
"""tools/extend_bench.py -- throughput of the GPU seed extension (cs_extend_batch_device, compseed_amd/csrc/extend.hip).

Workload: pairs shaped like the ones the reference extends for 150-bp reads -- query lengths, target lengths and h0 drawn from the pairs its own run
produced on the sorted150 golden set (tests/golden/bsw1/sorted150.default.bsw.npz), sequences synthetic: the target is the query with 0.5 %
substitutions and an occasional indel, followed by unrelated bases.  Pairs, sequences and results are resident in HBM; one launch per pass, band
w = 100 (the reference's first band try).  Reports pairs/s, DP cells/s inside the adaptive band, the oracle (CPU port of ksw_extend2) on the host's
cores next to it, and checks a sample of the results against that oracle.  bench.py calls run() for its `extension` side key; tools/profile_round.sh
runs this file under rocprofv3 --pmc for the VALU share of the kernel (this kernel is integer-compute-bound, not HBM-bound)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_pairs(n, dev, seed=5):
    """-> (pairs as a host array, qbuf, tbuf as uint8 tensors on `dev`)"""
    import torch
    import _oracle
    fx = _oracle.bsw_fixture("sorted150.default")
    g = torch.Generator(device=dev); g.manual_seed(seed)
    pick = torch.randint(0, fx["pairs"].size, (n,), device=dev, generator=g)
    fq = torch.from_numpy(fx["pairs"]["qlen"].astype(np.int64)).to(dev); ft = torch.from_numpy(fx["pairs"]["tlen"].astype(np.int64)).to(dev)
    fh = torch.from_numpy(fx["pairs"]["h0"].astype(np.int64)).to(dev)
    qlen = fq[pick]; tlen = torch.maximum(ft[pick], qlen); h0 = fh[pick]
    q_off = torch.zeros(n + 1, dtype=torch.int64, device=dev); q_off[1:] = torch.cumsum(qlen, 0)
    t_off = torch.zeros(n + 1, dtype=torch.int64, device=dev); t_off[1:] = torch.cumsum(tlen, 0)
    nq, nt = int(q_off[-1]), int(t_off[-1])
    qbuf = torch.randint(0, 4, (nq,), dtype=torch.uint8, device=dev, generator=g)
    tbuf = torch.randint(0, 4, (nt,), dtype=torch.uint8, device=dev, generator=g)
    # target prefix = the query, then substitutions; one pair in eight gets a 1..3-base deletion (the copy is shifted)
    owner = torch.repeat_interleave(torch.arange(n, device=dev), qlen)
    pos_in = torch.arange(nq, device=dev) - q_off[owner]
    shift = torch.where((owner % 8 == 0) & (pos_in > qlen[owner] // 2), 1 + owner % 3, torch.zeros_like(owner))
    src = torch.minimum(pos_in + shift, qlen[owner] - 1) + q_off[owner]
    tbuf[t_off[owner] + pos_in] = qbuf[src]
    del owner, pos_in, shift, src
    mut = torch.rand(nt, device=dev, generator=g) < 0.005
    tbuf = torch.where(mut, torch.randint(0, 4, (nt,), dtype=torch.uint8, device=dev, generator=g), tbuf)
    pairs = np.zeros(n, dtype=_oracle.BSW_PAIR_DT)
    pairs["q_off"], pairs["t_off"] = q_off[:-1].cpu().numpy(), t_off[:-1].cpu().numpy()
    pairs["qlen"], pairs["tlen"], pairs["h0"] = qlen.cpu().numpy(), tlen.cpu().numpy(), h0.cpu().numpy()
    return pairs, qbuf, tbuf


def run(n_pairs=2_000_000, steps=5, check=20000, cpu_pairs=200000, cpu_threads=16, device=0, lib=None, flags=0):
    import torch
    import compseed_amd as ca
    import _oracle
    dev = torch.device("cuda", device)
    pairs, d_q, d_t = make_pairs(n_pairs, dev)
    pr = np.zeros(n_pairs, dtype=ca.EXT_PAIR_DT)
    for f in ("q_off", "t_off", "qlen", "tlen", "h0"):
        pr[f] = pairs[f]
    d_p = torch.from_numpy(pr.view(np.uint8)).to(dev)
    d_o = torch.zeros(n_pairs * 24, dtype=torch.uint8, device=dev)
    q_bytes, t_bytes = d_q.numel(), d_t.numel()
    torch.cuda.synchronize()
    x = ca.Extender(device, ca.ExtParams(flags=flags)) if flags else ca.Extender(device)
    x.extend_device(d_p.data_ptr(), n_pairs, d_q.data_ptr(), q_bytes, d_t.data_ptr(), t_bytes, d_o.data_ptr(), 100)   # warm-up
    s0 = x.stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        x.extend_device(d_p.data_ptr(), n_pairs, d_q.data_ptr(), q_bytes, d_t.data_ptr(), t_bytes, d_o.data_ptr(), 100)
    wall = (time.perf_counter() - t0) / steps
    s1 = x.stats()
    kern_ms = (s1["kernel_ms"] - s0["kernel_ms"]) / steps
    cells = (s1["cells"] - s0["cells"]) / steps
    got = d_o.cpu().numpy().view(ca.EXT_RES_DT)
    # parity on a strided sample against the oracle (the checker)
    ids = np.arange(0, n_pairs, max(1, n_pairs // check))[:check]

    def host_subset(sel):  # the selected pairs with their sequences copied into compact host buffers
        sub = pairs[sel].copy()
        ql, tl = sub["qlen"].astype(np.int64), sub["tlen"].astype(np.int64)
        qo = np.zeros(sel.size + 1, np.int64); np.cumsum(ql, out=qo[1:])
        to = np.zeros(sel.size + 1, np.int64); np.cumsum(tl, out=to[1:])
        iq = np.repeat(sub["q_off"].astype(np.int64) - qo[:-1], ql) + np.arange(qo[-1])
        it = np.repeat(sub["t_off"].astype(np.int64) - to[:-1], tl) + np.arange(to[-1])
        hq = d_q[torch.from_numpy(iq).to(dev)].cpu().numpy(); ht = d_t[torch.from_numpy(it).to(dev)].cpu().numpy()
        sub["q_off"], sub["t_off"] = qo[:-1], to[:-1]
        return dict(mat=np.array(list(ca.ExtParams().mat), dtype=np.int8), pairs=sub, qbuf=hq, tbuf=ht,
                    meta=np.tile(np.array([[0, 100, 100, 5, 6, 1, 6, 1] + [0] * 9], dtype=np.int32), (sel.size, 1)))
    fx = host_subset(ids)
    want = _oracle.bsw_extend(fx, threads=cpu_threads)
    ok = bool(np.array_equal(got[ids], want.astype(ca.EXT_RES_DT)))
    out = {"pairs_per_s": n_pairs / (kern_ms * 1e-3), "pairs_per_s_wall": n_pairs / wall, "kernel_ms": kern_ms, "pairs": n_pairs,
           "dp_cells_per_s": cells / (kern_ms * 1e-3), "dp_cells_per_pair": cells / n_pairs, "mean_qlen": float(pairs["qlen"].mean()), "mean_tlen": float(pairs["tlen"].mean()),
           "band_w": 100, "bit_exact_vs_oracle": ok, "checked_pairs": int(ids.size),
           "workload": "pair shapes (qlen, tlen, h0) of the reference's own extensions on the sorted150 golden set, synthetic sequences (0.5 % substitutions, "
                       "1-3-base deletion in one pair of eight, unrelated tails); resident in HBM; mem_opt_init scoring"}
    if cpu_pairs > 0:
        k = min(cpu_pairs, n_pairs)
        fx = host_subset(np.arange(k))
        t0 = time.perf_counter()
        _oracle.bsw_extend(fx, threads=cpu_threads)
        dt = time.perf_counter() - t0
        out["cpu_port"] = {"pairs_per_s": k / dt, "threads": cpu_threads, "kind": "port", "sample": "first %d pairs, oracle/cs_bsw_oracle.c (scalar ksw_extend2 restated), %.1f s" % (k, dt)}
    # the REAL reference's extension code (its vectorised getScores8 / getScores16 and the scalar wrapper, untouched) timed inside the
    # reference's own program on the box's host cores: oracle/_ref/CompSeed.bswtrace with $CS_BSW_TIME sums the time of those calls
    # over all threads (oracle/ref_bsw_trace.cpp); reads = tests/golden/bsw1/indel150.txt against the golden reference
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "CompSeed.bswtrace")
    gold = os.path.join(ROOT, "tests", "golden")
    if cpu_pairs and os.path.exists(ref_bin) and os.path.exists(os.path.join(gold, "g1", "ref.bwt")):
        import re, subprocess
        try:
            env = dict(os.environ, CS_BSW_TIME="1"); env.pop("CS_BSW_TRACE", None)
            r = subprocess.run([ref_bin, "-t", str(cpu_threads), os.path.join(gold, "g1", "ref"), os.path.join(gold, "bsw1", "indel150.txt")], env=env, cwd="/tmp",
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=120)
            m = re.search(r"\[bswtrace\] pairs=(\d+) thread_seconds=([0-9.]+) cells=(\d+)", r.stderr)
            if m:
                pairs, ts, cells = int(m.group(1)), float(m.group(2)), int(m.group(3))
                out["cpu_reference"] = {"pairs_per_thread_second": pairs / ts, "threads": cpu_threads, "pairs_per_s_all_threads": pairs / ts * cpu_threads, "kind": "reference",
                                        "rect_cells_per_thread_second": cells / ts,
                                        "sample": "%d extensions the reference ran on 1200 reads with indels (w and 2w tries), its own AVX2 code, time summed inside its calls over %d threads" % (pairs, cpu_threads)}
        except Exception as ex:  # noqa: BLE001
            out["cpu_reference_error"] = repr(ex)
    x.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=2_000_000); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--cpu-pairs", type=int, default=200000); ap.add_argument("--flags", type=int, default=0, help="cs_ext_params_t.flags (CS_EXT_*)")
    a = ap.parse_args()
    print(json.dumps(run(a.pairs, a.steps, cpu_pairs=a.cpu_pairs, flags=a.flags)))
