for o in "" "--opt lep_arena_mb=16384" "--opt mem_cap=32" "--opt lep_arena_mb=16384 --opt mem_cap=32" "--opt lep_arena_mb=12288 --opt mem_cap=24"; do
python bench.py --steps 6 --warmup 2 --traffic none --cpu-seconds 0 --no-extension --side-workloads "" --no-host-io $o > gpurun_out/ab.json 2> gpurun_out/ab.err
python -c "
import json,sys
j=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print(sys.argv[1:], round(j['value']/1e6,1), round(j['ms_per_step'],2), round(j['one_step_at_a_time']['ms_per_step'],2), round(j['roofline']['kernel_ms_per_launch'],2), j['roofline']['overflow_mems_per_step'], j['parity']['bit_exact_vs_oracle'])
" $o
done
