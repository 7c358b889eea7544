R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; ulimit -c 0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rep -- python3 $R/bench.py --profile repeat50 --steps 2 --warmup 1 --in-flight 1 --cpu-seconds 0 --no-host-io --traffic none --no-extension --side-workloads "" > $R/gpurun_out/rep_prof.json 2> /tmp/prof_rep.err
f=$(find /tmp/prof_rep -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1])))
print(rows[0])
for r in rows[1:16]:
    r=list(r); r[0]=r[0][:60]; print(r)
PY
