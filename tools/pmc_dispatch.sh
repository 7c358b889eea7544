#!/bin/bash
# tools/pmc_dispatch.sh <tag> <pmc counters...> -- like pmc.sh but prints the counters of the first dispatches of the SMEM kernels one by one
tag=$1; shift
out=/tmp/pmcd_$tag
cd /tmp && export TMPDIR=/tmp; ulimit -c 0
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py ${CS_PMC_BENCH_ARGS:---reads 4000000 --steps 1 --warmup 0 --cpu-seconds 0 --check-reads 1000} > $out.log 2> $out.err
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.OrderedDict()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fwd_kernel" in k or "bwd_all" in k:
            key = (int(r["Dispatch_Id"]), "fwd" if "fwd_kernel" in k else "bwd")
            rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for i, (key, d) in enumerate(sorted(rows.items())):
    if i >= 8: break
    print(key, " ".join("%s=%.3g" % (c, v) for c, v in sorted(d.items())))
PY
rm -rf $out
