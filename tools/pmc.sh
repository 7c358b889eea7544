#!/bin/bash
# tools/pmc.sh <tag> <pmc counters...> -- one rocprofv3 counter pass over a short bench run (run on the GPU box via gpurun)
# Counters go in their own run with no tracing flags other than --kernel-trace (gpurun refuses --pmc with sys/hip traces).
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py ${CS_PMC_BENCH_ARGS:---reads 4000000 --steps 1 --warmup 0 --cpu-seconds 0 --check-reads 1000} > $out.log 2> $out.err
python3 - "$out" > $out.summary.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if any(x in k for x in ("smem_kernel", "sal_walk", "fwd_kernel", "fwd0_kernel", "bwd_all", "bwd_win", "r2text", "r3text", "sort_compact", "sal_gather")):
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            print("%-62s %-28s %18.0f  (%d dispatches)" % (k, c, v, calls[(k, c)]))
PY
rm -rf $out   # raw csv is large; gpurun_out is capped at 64 MiB
cat $out.summary.txt
