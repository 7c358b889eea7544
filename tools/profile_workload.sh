#!/bin/bash
# tools/profile_workload.sh <tag> <profile> -- the kernel-level picture of a side workload (tools/synth.py profile, e.g. repeat50), one
# pass at a time: rocprofv3 --kernel-trace --stats (per-kernel time, stage span) and, in a run of its own, the SQ counters per kernel.
#   gpurun --timeout 900 -- 'tools/profile_workload.sh r03_repeat50 repeat50'
tag=${1:-rXX}; prof=${2:-repeat50}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; ulimit -c 0
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $R/bench.py --profile $prof --steps 2 --warmup 1 --in-flight 1 --cpu-seconds 0 --no-host-io --traffic none --no-extension --side-workloads "" > $O/${tag}_bench_profiled.json 2> /tmp/prof_$tag.err || { tail -5 /tmp/prof_$tag.err; exit 1; }
python3 - "$tag" <<'PY'
import csv, glob, sys, os, json
tag = sys.argv[1]; R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob("/tmp/prof_%s/**/*kernel_stats.csv" % tag, recursive=True)[0]
rows = list(csv.reader(open(f)))
keep = [rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in ("csd::", "anonymous namespace", "rocprim", "max_len", "collect_overflow", "patch_counts", "sort_compact", "_words_kernel", "pack_"))]
with open(os.path.join(R, "gpurun_out", tag + "_kernel_stats.csv"), "w", newline="") as o:
    w = csv.writer(o)
    for r in keep:
        r = list(r); r[0] = r[0][:110]; w.writerow(r)
STAGE = ("fwd_kernel", "fwd0_kernel", "bwd_all_kernel", "bwd_wide_kernel", "bwd_win", "r2text_kernel", "r3text_kernel", "init_tasks_kernel")
f = glob.glob("/tmp/prof_%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in STAGE)]
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
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/sq_$tag -- python3 $R/bench.py --profile $prof --reads 4000000 --steps 1 --warmup 0 --in-flight 1 --cpu-seconds 0 --no-host-io --traffic none --check-reads 1000 --no-extension --side-workloads "" > /tmp/sq_$tag.json 2> /tmp/sq_$tag.err || { tail -5 /tmp/sq_$tag.err; exit 1; }
python3 - "$tag" > $O/${tag}_sq_counters.txt <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob("/tmp/sq_%s/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].split("::")[-1]
        if any(x in k for x in ("smem_kernel", "fwd_kernel", "fwd0_kernel", "bwd_all", "bwd_win", "bwd_wide", "r2text", "r3text", "sort_compact", "sal_")):
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items()):
    for c, v in sorted(d.items()):
        print("%-28s %-22s %18.0f  (%d dispatches)" % (k, c, v, calls[(k, c)]))
PY
cat $O/${tag}_kernel_stats.csv | cut -c1-200
