#!/bin/bash
# tools/r06_session4.sh [tag] -- new m-estimators on the device, poll statistics of the dataflow waits (is a wait that ends on the RMW poll a stale line?)
out=gpurun_out/${1:-r06e}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_gtsam_shim.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5 > $out/tests.log; tail -3 $out/tests.log
timeout 200 python tools/df_stress.py 60 3 sphere2500 > $out/stress_sphere2500.txt 2> $out/stress.err; tail -1 $out/stress_sphere2500.txt
timeout 200 python tools/df_stress.py 60 1 sphere2500 > $out/stress_sphere2500_1thread.txt 2>> $out/stress.err; tail -1 $out/stress_sphere2500_1thread.txt
timeout 200 python tools/df_stress.py 60 3 bal300 > $out/stress_bal300.txt 2>> $out/stress.err; tail -1 $out/stress_bal300.txt
timeout 600 python bench.py --workloads off --cpu-baseline off --traffic off --skip-dense-roofline > $out/bench_quick.json 2> $out/bench.err
python - <<PY
import json
j=json.loads([l for l in open('$out/bench_quick.json') if l.startswith('{')][-1])
print('value', j['value'], 'chol ms', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'], 'phases', j['phase_ms_per_call'])
PY
