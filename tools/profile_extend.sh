#!/bin/bash
# tools/profile_extend.sh <tag> -- the seed-extension kernel's numbers for profiles/: throughput line, kernel trace stats, SQ counters (two passes).
#   gpurun --timeout 600 -- 'tools/profile_extend.sh r03_extend'
tag=${1:-rXX_extend}
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; ulimit -c 0
O=$R/gpurun_out
python3 $R/tools/extend_bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err || { tail -5 $O/${tag}_bench.err; exit 1; }
cat $O/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_$tag -- python3 $R/tools/extend_bench.py --steps 3 --cpu-pairs 0 > /dev/null 2> /tmp/px_$tag.err
f=$(find /tmp/px_$tag -name '*kernel_stats.csv' | head -1); head -1 $f > $O/${tag}_kernel_stats.csv; grep -E "extend_kernel|extend16_kernel|extend_lanes_kernel|lanes_keys|max_qlen|rocprim" $f >> $O/${tag}_kernel_stats.csv
for pass in A B C; do
  if [ $pass = A ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; elif [ $pass = B ]; then C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVES"; else C="SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pq_${tag}_$pass -- python3 $R/tools/extend_bench.py --steps 1 --cpu-pairs 0 > /dev/null 2> /tmp/pq_$tag.err
done
python3 - "$tag" > $O/${tag}_sq_counters.txt <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("/tmp/pq_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        if "extend_kernel" in r["Kernel_Name"] or "extend16_kernel" in r["Kernel_Name"] or "extend_lanes_kernel" in r["Kernel_Name"]:   # the kernels of one call, summed
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c, v in sorted(agg.items()):
    print("extend*_kernel %-22s %18.0f  (%d dispatches)" % (c, v, n[c]))
if agg.get("SQ_WAVE_CYCLES"):
    print("VALU busy per wave cycle  %.3f   waiting share %.3f   (x resident waves per SIMD = share of a SIMD's issue slots)" % (agg["SQ_ACTIVE_INST_VALU"] / agg["SQ_WAVE_CYCLES"], agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"]))
if agg.get("SQ_THREAD_CYCLES_VALU") and agg.get("SQ_ACTIVE_INST_VALU"):
    print("lanes active per VALU cycle  %.3f of 64" % (agg["SQ_THREAD_CYCLES_VALU"] / agg["SQ_ACTIVE_INST_VALU"] / 4.0))
if agg.get("SQ_BUSY_CYCLES"):
    print("VALU active cycles / SQ busy cycles  %.3f" % (agg["SQ_ACTIVE_INST_VALU"] / agg["SQ_BUSY_CYCLES"]))
PY
cat $O/${tag}_sq_counters.txt $O/${tag}_kernel_stats.csv
rm -rf /tmp/px_$tag /tmp/pq_${tag}_*
