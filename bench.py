#!/usr/bin/env python3
"""bench.py -- reads/sec of the SMEM seeding hot path on N MI355X (one process per GPU).

A "step" is one pass of the hot path (ASCII->nt4, 3-round SMEM collection, sort, SAL) over one batch of synthetic
150-bp reordered reads that is already resident in HBM; results stay in HBM.  Every rank holds a replica of the
index and its own contiguous share of the read run (weak scaling, no data-path collective: the path shards by
reads).  Rank 0 prints ONE JSON line.  See DESIGN.md "Measurement" for the roofline and cpu_baseline definitions.

    python bench.py                       # 1 GPU, defaults
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("CS_BENCH_GENOME_MBP", "3100")),
                    help="synthetic genome size in Mbp (3100 = hg19 scale)")
    ap.add_argument("--reads", type=int, default=int(os.environ.get("CS_BENCH_READS", "10000000")), help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--no-sal", action="store_true")
    ap.add_argument("--host-io", action="store_true", help="also time the host-buffer variant (PCIe in and out) after the timed region")
    ap.add_argument("--sst", type=int, default=int(os.environ.get("CS_BENCH_SST", "1")), help="on-device SST memo (1 = on, 0 = off)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample (0 = skip)")
    ap.add_argument("--check-reads", type=int, default=20000, help="reads compared bit-for-bit against the oracle after the timed region")
    ap.add_argument("-k", type=int, default=19); ap.add_argument("-r", type=float, default=1.5)
    ap.add_argument("-y", type=int, default=20); ap.add_argument("-c", type=int, default=500); ap.add_argument("-s", type=int, default=10)
    args = ap.parse_args()

    import torch
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the seeding engine has no CPU path")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import compseed_amd as ca
    from compseed_amd.sharding import Dist
    import synth
    D = Dist("nccl")                      # RCCL; used for the barrier and the max-over-ranks clock only
    rank, world = D.rank, D.world
    if world != args.gpus and world > 1:
        log("WORLD_SIZE %d != --gpus %d, using WORLD_SIZE" % (world, args.gpus))

    # ---- workload: synthetic genome (same on every rank), index built on this rank's GPU, reads of this rank's share
    t0 = time.time()
    L = int(args.genome_mbp * 1e6)
    G = synth.make_genome(L, seed=20261003, device=dev)
    g_host = G.cpu().numpy()
    torch.cuda.synchronize()
    t1 = time.time()
    torch.cuda.empty_cache()
    ix = ca.Index.build(g_host, local)
    t2 = time.time()
    eng = ca.Engine(ix, local)
    t3 = time.time()
    bases, off = synth.make_reads(G, args.reads, args.read_len, seed=777 + rank, p_sub=0.005, sort=True,
                                  lo_frac=rank / world, hi_frac=(rank + 1) / world)
    n_bases = bases.numel()
    torch.cuda.synchronize()
    del G
    torch.cuda.empty_cache()
    if rank == 0:
        log("genome %.0f Mbp: generate %.1fs, index build %.1fs (seq_len %d), upload %.1fs, reads %d x %d in %.1fs" %
            (args.genome_mbp, t1 - t0, t2 - t1, ix.view.seq_len, t3 - t2, args.reads, args.read_len, time.time() - t3))
    par = ca.Params(k=args.k, r=args.r, s=args.s, c=args.c, y=args.y, want_sal=0 if args.no_sal else 1, sst_mode=args.sst)

    def step():
        return eng.seed_batch_device(bases.data_ptr(), off.data_ptr(), args.reads, n_bases, par)

    if args.warmup == 0:
        step()  # never time the first call of an engine: it sizes and allocates the batch buffers (setup, like the index upload)
    for _ in range(args.warmup):
        res = step()
    eng.reset_stats()
    D.barrier()
    ts = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - ts)
    st = eng.stats()

    total_reads = args.reads * world * args.steps
    value = total_reads / elapsed
    out = {
        "metric": "reads/sec (150bp) seeded", "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "synthetic %.0f Mbp genome (Alu-like family 10%%, segdups, tandem arrays), %d x %d bp position-sorted "
                               "reads per GPU per step, 0.5%% substitutions, -k %d -r %g -y %d -c %d%s" %
                               (args.genome_mbp, args.reads, args.read_len, args.k, args.r, args.y, args.c, " (mems only)" if args.no_sal else " + SAL"),
                   "genome_mbp": args.genome_mbp, "reads_per_gpu_per_step": args.reads, "read_len": args.read_len,
                   "parallelism": "reads sharded over %d GPU(s), index replicated" % world},
    }

    if rank == 0:
        # ---- parity gate + algorithmic bytes + CPU baseline, all on a bounded prefix of rank 0's reads (oracle = checker)
        import _oracle
        bwt_words, sa = ix.arrays()
        oidx = _oracle.OracleIndex.from_arrays(ix.view.primary, [ix.view.L2[i] for i in range(1, 5)], bwt_words, sa, 32)
        opar = _oracle.make_params(k=args.k, r=args.r, s=args.s, c=args.c, y=args.y)
        nchk = min(args.check_reads, args.reads)
        hb = bases[: nchk * args.read_len].cpu().numpy(); ho = off[: nchk + 1].cpu().numpy().astype(np.uint64)
        want = oidx.seed_batch(hb, ho, opar, mode=1, sst_batch=512, want_sal=not args.no_sal, threads=min(16, len(os.sched_getaffinity(0))))
        mo = eng.download(res.ptr["mem_off"], np.uint64, nchk + 1)
        nm = int(mo[-1])
        mm = eng.download(res.ptr["mems"], ca.INTV_DT, nm)
        ok = np.array_equal(mo, want["mem_off"]) and np.array_equal(mm, want["mems"])
        if not args.no_sal:
            so = eng.download(res.ptr["seed_off"], np.uint64, nchk + 1)
            ss = eng.download(res.ptr["seeds"], ca.SEED_DT, int(so[-1]))
            ok = ok and np.array_equal(so, want["seed_off"]) and np.array_equal(ss, want["seeds"])
        out["parity"] = {"checked_reads": nchk, "bit_exact_vs_oracle": bool(ok)}
        if not ok:
            log("PARITY FAILURE against the oracle on the first %d reads" % nchk)
        ws = want["stats"]
        # algorithmic bytes of the SMEM kernel per read: 64 B x Occ blocks of the REAL bwt_extend calls under the
        # reference's cache policy (fresh SSTs per 512 reads) + the read itself + 32 B per mem written
        unc = oidx.seed_batch(hb, ho, opar, mode=0, want_sal=False, threads=min(16, len(os.sched_getaffinity(0))))["stats"]
        per_read = (64.0 * ws["bwt_blocks"] + 32.0 * ws["n_mems"]) / nchk + args.read_len
        per_read_unc = (64.0 * unc["bwt_blocks_uncached"] + 32.0 * ws["n_mems"]) / nchk + args.read_len
        # The SMEM stage is the dominant device work.  In the default (split) mode one pass over a batch = a short chain of
        # fwd_kernel / bwd_all_kernel launches; their summed duration per pass is measured with HIP events on the engine's
        # stream (first launch to last) and must agree with rocprofv3's TotalDurationNs(fwd_kernel)+(bwd_all_kernel) per pass.
        launches = max(1, st["seed_kernel_launches"])          # passes of the stage (1 per step unless the batch is split)
        kern_ms = st["seed_kernel_ms"] / launches
        reads_per_launch = args.reads * args.steps / launches
        achieved = per_read * reads_per_launch / (kern_ms * 1e-3) / 1e9
        fused = os.environ.get("CS_SMEM_MODE", "") == "fused"
        pmc = None
        try:  # HBM bytes per pass from the committed PMC run of this workload (tools/pmc.sh), if there is one
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = "%g_%d_%d" % (args.genome_mbp, args.reads, args.read_len)
            pmc = pm.get(key, {}).get("smem_stage_bytes_per_pass")
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": pmc, "kernel": "smem_kernel" if fused else "SMEM stage = all fwd0 / fwd / bwd_win / bwd_win0 / bwd_wide / r2text / r3text launches of one pass",
                           "traffic_GBps": (pmc / (kern_ms * 1e-3) / 1e9) if pmc else None,
                           "traffic_frac": (pmc / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pmc else None,
                           "note": "achieved/frac use SURVEY 8(d)'s algorithmic bytes = what the REFERENCE's bwt_extend calls touch; the kernels answer most of "
                                   "those calls from a jump table and text-side arrays (DESIGN.md 4.2), so real traffic is several times lower and frac can exceed 1; "
                                   "traffic_frac is the fetched bytes over the same time over the same peak",
                           "kernel_ms_per_launch": kern_ms, "launches": launches,
                           "algorithmic_bytes_per_read": per_read, "uncached_bytes_per_read": per_read_unc,
                           "bwt_extend_queries_per_read": st["bwt_queries"] / (args.reads * args.steps),
                           "device_sst_hit_rate": 1.0 - st["bwt_calls"] / max(1, st["bwt_queries"]),
                           "reference_sst_hit_rate": 1.0 - ws["bwt_calls"] / max(1, ws["bwt_queries"]),
                           "reference_bwt_extend_queries_per_read": ws["bwt_queries"] / nchk,
                           "round3_seeds_from_text_per_read": st["r3_text_seeds"] / (args.reads * args.steps),
                           "sweeps_read_off_the_text_per_read": st["sweep_text_calls"] / (args.reads * args.steps),
                           "reseed_calls_answered_from_text_per_read": st["reseed_text_calls"] / (args.reads * args.steps),
                           "reseed_calls_of_unique_smems_left_to_index_per_read": st["reseed_index_calls"] / (args.reads * args.steps),
                           "sal_ms_per_step": st["sal_kernel_ms"] / args.steps, "total_gpu_ms_per_step": st["total_ms"] / args.steps,
                           "overflow_reads_per_step": st["overflow_reads"] / args.steps,
                           "overflow_pass_ms_per_step": st["overflow_kernel_ms"] / args.steps}
        if args.cpu_seconds > 0:
            cores = min(len(os.sched_getaffinity(0)), 16 * max(1, torch.cuda.device_count()))  # the box's CPU share: 16 per GPU
            probe_n = min(nchk, 20000)
            tp = time.perf_counter()
            oidx.seed_batch(hb[: probe_n * args.read_len], ho[: probe_n + 1], opar, mode=1, want_sal=not args.no_sal, threads=cores)
            rate = probe_n / max(1e-6, time.perf_counter() - tp)
            ncpu = int(min(args.reads, max(probe_n, rate * args.cpu_seconds)))
            cb = bases[: ncpu * args.read_len].cpu().numpy(); co = off[: ncpu + 1].cpu().numpy().astype(np.uint64)
            tp = time.perf_counter()
            oidx.seed_batch(cb, co, opar, mode=1, sst_batch=512, want_sal=not args.no_sal, threads=cores)
            dt = time.perf_counter() - tp
            out["cpu_baseline"] = {"value": ncpu / dt, "unit": "reads/s", "cores": cores, "kind": "port",
                                   "sample": "first %d reads of rank 0's batch, oracle/cs_oracle.c in CompSeed mode (SST per 512 reads), %d threads, %.1f s"
                                             % (ncpu, cores, dt)}
        if args.host_io:  # the boundary's host-buffer form: reads from pageable host memory, results into pinned host memory
            hb_all = bases.cpu().numpy(); ho_all = off.cpu().numpy().astype(np.uint64)
            eng.seed_batch(hb_all, ho_all, par, copy=False)  # warm-up: pinned result buffers are allocated on first use
            tp = time.perf_counter()
            eng.seed_batch(hb_all, ho_all, par, copy=False)
            dt = time.perf_counter() - tp
            out["pcie_inclusive"] = {"reads_per_s": args.reads / dt, "ms_per_step": 1e3 * dt,
                                     "note": "cs_engine_seed_batch: H2D of the reads + D2H of mems and seeds, not overlapped; never `value`"}
        print(json.dumps(out), flush=True)
    D.close()
    eng.close(); ix.close()


if __name__ == "__main__":
    main()
