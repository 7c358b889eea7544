#!/usr/bin/env python3
"""tools/kernel_table.py <bench.json> <kernel_stats.csv> <sq_counters.txt> -- the per-kernel table of a round (profiles/<tag>_kernel_table.md):
model bytes (cs_engine_traffic_model, counted on the device), FETCH_SIZE (rocprofv3 --pmc child of bench.py), time with the dispatches serialised
by the profiler and in the normal, overlapped run (rocprofv3 --kernel-trace --stats of a 3-pass run), fraction of the 8 TB/s peak, and the share
of a SIMD's issue slots spent on VALU instructions (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x resident waves per SIMD)."""
import csv
import json
import sys

bench, stats, sq = sys.argv[1:4]
d = json.load(open(bench))
r = d["roofline"]
passes = 3.0  # steps 2 + warmup 1 in tools/profile_round.sh
ms_overlapped = {}
for row in csv.DictReader(open(stats)):
    name = row["Name"].replace("void ", "").split("(")[0].split("<")[0].split("::")[-1]
    ms_overlapped[name] = ms_overlapped.get(name, 0.0) + float(row["TotalDurationNs"]) / 1e6 / passes
sqc = {}
for ln in open(sq):
    f = ln.split()
    if len(f) >= 3:
        sqc.setdefault(f[0], {})[f[1]] = float(f[2])
WAVES = {"fwd0_kernel": 8, "fwd_kernel": 6, "bwd_win_kernel": 6, "bwd_win0_kernel": 6, "bwd_wide_kernel": 6, "bwd_all_kernel": 5, "r2text_kernel": 5, "r3text_kernel": 5}
print("| kernel | dispatches / pass | model bytes / pass | FETCH_SIZE / pass | fetch / model | ms serialised | ms in the overlapped run | fetch at serialised time, of 8 TB/s | VALU share of issue slots | waiting (SQ_WAIT_ANY / wave cycles) |")
print("|---|---|---|---|---|---|---|---|---|---|")
tm = tf = 0.0
for name, k in sorted(r["kernels"].items(), key=lambda kv: -(kv[1].get("fetch_size_bytes") or 0)):
    mb, fb, ms = k.get("model_bytes", 0), k.get("fetch_size_bytes"), k.get("ms_serialised")
    tm += mb; tf += fb or 0
    c = sqc.get(name, {})
    valu = c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"] * WAVES.get(name, 4) if c.get("SQ_WAVE_CYCLES") else None
    wait = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
    print("| `%s` | %s | %.2f GB | %s | %s | %s | %s | %s | %s | %s |" % (
        name, "%.0f" % k["dispatches"] if k.get("dispatches") else "-", mb / 1e9, "%.2f GB" % (fb / 1e9) if fb else "-",
        "%.2f" % (fb / mb) if fb and mb else "-", "%.2f" % ms if ms else "-", "%.2f" % ms_overlapped[name] if name in ms_overlapped else "-",
        "%.3f" % (fb / (ms * 1e-3) / 8e12) if fb and ms else "-", "%.2f" % valu if valu is not None else "-", "%.2f" % wait if wait is not None else "-"))
print("| **SMEM stage** | %d launches | **%.2f GB** (+ %.2f GB streamed) | **%.2f GB** | %.2f | - | **%.2f** (HIP events, live) | **%.3f** (model: %.3f) | | |" % (
    sum(int(k.get("dispatches") or 0) for k in r["kernels"].values()), tm / 1e9, r["model_stream_bytes_per_launch"] / 1e9, tf / 1e9,
    tf / r["model_bytes_per_launch"], r["kernel_ms_per_launch"], r["traffic_frac"] or 0, r["frac"]))
print()
print("reads/s %.1f M, ms/step %.2f; PCIe-inclusive %s; counter calibration: %s B of FETCH_SIZE per random 32-byte record read." % (
    (d["value"] or 0) / 1e6, d["ms_per_step"], json.dumps({k: round(v / 1e6, 1) for k, v in d.get("pcie_inclusive", {}).items() if k.endswith("reads_per_s")}),
    r["traffic_calibration"]["fetch_size_bytes_per_random_32B_record_read"] if r.get("traffic_calibration") else "-"))
