#!/bin/bash
# tools/profile_round.sh <tag> -- rocprofv3 kernel-trace summary (+ optional FETCH_SIZE pass) of the default bench command.
# Run on the GPU box:  gpurun -- 'tools/profile_round.sh r01_v2'
# Writes small summaries to gpurun_out/ (copy the ones to be judged into profiles/).
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; ulimit -c 0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 > $R/gpurun_out/${tag}_bench_profiled.json 2> /tmp/prof_$tag.err
python3 - "$tag" <<'PY'
import csv, glob, sys, os
tag = sys.argv[1]; R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob("/tmp/prof_%s/**/*kernel_stats.csv" % tag, recursive=True)[0]
rows = list(csv.reader(open(f)))
keep = [rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in ("csd::", "anonymous namespace", "rocprim", "max_len", "collect_overflow", "patch_counts", "sort_compact"))]
with open(os.path.join(R, "gpurun_out", tag + "_kernel_stats.csv"), "w", newline="") as o:
    w = csv.writer(o)
    for r in keep:
        r = list(r); r[0] = r[0][:110]; w.writerow(r)
# wall span of the SMEM stage per pass from the kernel trace: first fwd_kernel start to last fwd/bwd kernel end.  (The round-3
# fwd_kernel launch runs beside the first forward launch and bwd_wide_kernel beside bwd_all_kernel on their own streams, so
# the summed durations exceed the span.)
import json
f = glob.glob("/tmp/prof_%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))
      if any(k in r["Kernel_Name"] for k in ("fwd_kernel", "fwd0_kernel", "bwd_all_kernel", "bwd_wide_kernel", "bwd_win", "r2text_kernel", "r3text_kernel", "init_tasks_kernel"))]
ev.sort()
passes, cur, launches = [], None, []
for s0, e0, name in ev:
    if "init_tasks" in name:
        if cur: passes.append(cur)
        cur = [None, None, 0, 0.0]; launches = []
        continue
    if cur is None: continue
    launches.append([name.split("(")[0].split("::")[-1][:16], round((s0 - (cur[0] or s0)) / 1e6, 3), round((e0 - s0) / 1e6, 3)])
    cur[0] = s0 if cur[0] is None else min(cur[0], s0); cur[1] = e0 if cur[1] is None else max(cur[1], e0); cur[2] += 1; cur[3] += (e0 - s0) / 1e6
if cur: passes.append(cur)
out = [{"span_ms": (p[1] - p[0]) / 1e6, "launches": p[2], "summed_kernel_ms": p[3]} for p in passes if p[0] is not None]
json.dump({"passes": out, "last_pass_launches_name_startms_durms": launches}, open(os.path.join(R, "gpurun_out", tag + "_stage_span.json"), "w"), indent=1)
print(json.dumps(out))
PY
if [ -n "$2" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$tag -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --check-reads 1000 > /tmp/pmc_$tag.json 2> /tmp/pmc_$tag.err
  python3 - "$tag" <<'PY'
import csv, glob, sys, os, collections, json
tag = sys.argv[1]; R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(float); n = collections.Counter(); per = []
for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if r["Counter_Name"] == "FETCH_SIZE" and any(t in k for t in ("fwd_kernel", "fwd0_kernel", "bwd_", "r2text", "r3text", "sal_", "sort_compact")):
            per.append([int(r.get("Dispatch_Id", 0)), k.split("(")[0].split("::")[-1][:18], round(float(r["Counter_Value"]) / 1048576, 3)])
        name = "fwd0_kernel" if "fwd0_kernel" in k else "fwd_kernel" if "fwd_kernel" in k else "bwd_win0_kernel" if "bwd_win0" in k else "bwd_win_kernel" if "bwd_win" in k else "r2text_kernel" if "r2text" in k else "r3text_kernel" if "r3text" in k else "bwd_all_kernel" if "bwd_all" in k else "bwd_wide_kernel" if "bwd_wide" in k else "smem_kernel" if "smem_kernel" in k else "sal_gather" if "sal_gather" in k else None
        if name and r["Counter_Name"] == "FETCH_SIZE":
            agg[name] += float(r["Counter_Value"]); n[name] += 1
out = {k: {"FETCH_SIZE_KiB_sum": v, "dispatches": n[k]} for k, v in agg.items()}
per.sort()
out["passes_in_this_run"] = max(1, n.get("r3text_kernel", 0) or n.get("fwd0_kernel", 0) or 1)  # bench.py --warmup 0 adds an untimed priming pass
out["per_dispatch_id_kernel_GiB"] = per
json.dump(out, open(os.path.join(R, "gpurun_out", tag + "_pmc_fetch.json"), "w"), indent=1)
print(json.dumps(out))
PY
fi
