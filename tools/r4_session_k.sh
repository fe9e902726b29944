#!/bin/bash
out=gpurun_out/r4k; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $out/gpu_tests.log
grep -h "passed\|failed" $out/gpu_tests.log
for w in sphere2500 w20000; do
  timeout 300 python bench.py --workload $w --cpu-baseline off --skip-dense-roofline --traffic off > $out/bench_$w.json 2> $out/bench_$w.err
done
python - <<PY
import json
for w in ('sphere2500','w20000'):
    j=json.load(open('$out/bench_'+w+'.json')); print(w, round(j['value'],1), 'chol', round(j['phase_ms_per_call']['cholesky'],3), 'ms/step', round(j['ms_per_step'],3), 'ttc', j['time_to_converged_python_mirror_warm_s'], 'err', j['converged_error'], 'mem', j['device_memory_per_handle_bytes'])
PY
