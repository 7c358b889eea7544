#!/bin/bash
# tools/timeline.sh <tag> -- kernel trace of two bench passes; prints every launch of the SMEM stage of the last pass (start ms, duration ms)
tag=${1:-tl}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; ulimit -c 0
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-host-io --traffic none ${@:2} > /dev/null 2> /tmp/tl_$tag.err
python3 - "$tag" <<'PY'
import csv, glob, sys, os
tag = sys.argv[1]; R = os.environ["GRAFT_REPO_ROOT"]
STAGE = ("fwd_kernel", "fwd0_kernel", "bwd_all_kernel", "bwd_wide_kernel", "bwd_win", "r2text_kernel", "r3text_kernel", "init_tasks_kernel")
f = glob.glob("/tmp/tl_%s/**/*kernel_trace.csv" % tag, recursive=True)[0]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)) if any(k in r["Kernel_Name"] for k in STAGE)]
ev.sort()
passes, cur = [], None
for s0, e0, name in ev:
    if "init_tasks" in name:
        if cur: passes.append(cur)
        cur = []
        continue
    if cur is not None: cur.append((s0, e0, name.split("(")[0].split("::")[-1][:18]))
if cur: passes.append(cur)
p = [q for q in passes if q and not any(", t" in n or "<t" in n for _, _, n in q)][-1]   # last timed pass (not the counting one)
t0 = p[0][0]
with open(os.path.join(R, "gpurun_out", tag + "_timeline.txt"), "w") as o:
    for s0, e0, n in p: o.write("%-20s %8.3f %8.3f\n" % (n, (s0 - t0) / 1e6, (e0 - s0) / 1e6))
    o.write("span %.3f\n" % ((max(e for _, e, _ in p) - t0) / 1e6))
PY
