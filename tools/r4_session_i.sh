#!/bin/bash
out=gpurun_out/r4i; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q 2>&1 | tail -6 > $out/gpu_tests.log; tail -2 $out/gpu_tests.log
timeout 300 python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $out/bench.json 2> $out/bench.err
for w in w20000 sphere2500; do for d in 3 4; do
  GTG_ND_DEPTH=$d timeout 300 python bench.py --workload $w --steps 12 --warmup 3 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $out/bench_${w}_d$d.json 2> $out/bench_${w}_d$d.err
done; done
GTG_ND_DEPTH=4 timeout 200 python tools/df_trace_pose.py w20000 > $out/trace_w20000_d4.json 2> $out/trace.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], round(j['value'],1), 'chol', round(j['phase_ms_per_call']['cholesky'],3), 'ms/step', round(j['ms_per_step'],3), 'err', j['converged_error'], 'mem', j['device_memory_per_handle_bytes'])
    except Exception as e: print(f,'failed',e)
j=json.load(open('$out/trace_w20000_d4.json')); print('trace total', j['total_us'], 'slow PD', j['slowest_pd_final_pieces(us, J, pieces, steps)'][-2:], 'last', j['last_tiles_out'][-3:])
PY
