for o in "" "--opt stagger=1" "--opt stagger=1 --opt r3_text_iter=3" "--opt stagger=1 --opt r3_text_iter=5"; do
python bench.py --steps 8 --warmup 2 --traffic none --cpu-seconds 0 --no-extension --side-workloads "" --no-host-io $o > gpurun_out/ab.json 2> gpurun_out/ab.err
python -c "
import json,sys
j=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print(sys.argv[1:], round(j['value']/1e6,1), round(j['ms_per_step'],2), round(j['one_step_at_a_time']['ms_per_step'],2), j['parity']['bit_exact_vs_oracle'])
" $o
done
