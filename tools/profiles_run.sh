#!/bin/bash
# tools/profiles_run.sh <tag> -- workload sensitivity: the default bench command on every workload profile of
# tools/synth.py, plus BASELINE configs[4]'s parameters (-r 1.0 -y 20) on the default one.  One JSON line per run, collected into
# gpurun_out/<tag>_workload_profiles.jsonl (copy into profiles/).  Run on the GPU box:  gpurun -- 'tools/profiles_run.sh r02'
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/${tag}_workload_profiles.jsonl
: > $out
for p in default repeat50 err1 err2indel shuffled repeat50err1; do
  echo "[profiles] $p" >&2
  timeout -k 10 400 python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-host-io --profile $p >> $out 2>> $R/gpurun_out/${tag}_workload_profiles.err || echo "{\"profile\": \"$p\", \"failed\": true}" >> $out
done
echo "[profiles] default -r 1.0 -y 20" >&2
timeout -k 10 400 python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-host-io -r 1.0 -y 20 >> $out 2>> $R/gpurun_out/${tag}_workload_profiles.err || echo "{\"profile\": \"default -r 1.0\", \"failed\": true}" >> $out
python3 - "$out" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if d.get("failed"):
        print(d); continue
    r = d["roofline"]
    print("%-14s %-14s %7.1f M reads/s  stage %6.1f ms  model %5.0f B/read  fetch %5.0f B/read  frac %.3f  no-FM %.2f  r3text %.1f  r2text %.2f  sweeps %.2f  mems %.1f seeds %.1f  parity %s" % (
        d["config"]["profile"], "-r %g -y %s" % (float(d["config"]["workload"].split("-r ")[1].split()[0]), d["config"]["workload"].split("-y ")[1].split()[0]),
        (d["value"] or 0) / 1e6, r["kernel_ms_per_launch"], r["model_bytes_per_read"], (r["traffic"] or 0) / d["config"]["reads_per_gpu_per_step"], r["frac"],
        r["answered_without_fm_index"], r["round3_seeds_from_text_per_read"], r["reseed_calls_answered_from_text_per_read"], r["sweeps_read_off_the_text_per_read"],
        r["mems_per_read"], r["seeds_per_read"], d["parity"]["bit_exact_vs_oracle"]))
PY
