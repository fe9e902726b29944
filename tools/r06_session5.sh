#!/bin/bash
# tools/r06_session5.sh [tag] -- single flag words (no shadow), full GPU suite, stress, quick bench
out=gpurun_out/${1:-r06g}; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $out/gpu_tests.log; tail -3 $out/gpu_tests.log
timeout 200 python tools/df_stress.py 60 3 sphere2500 > $out/stress_sphere2500.txt 2> $out/stress.err; tail -1 $out/stress_sphere2500.txt
timeout 200 python tools/df_stress.py 60 3 bal300 > $out/stress_bal300.txt 2>> $out/stress.err; tail -1 $out/stress_bal300.txt
for r in 1 2; do timeout 600 python bench.py --workloads off --cpu-baseline off --traffic off --skip-dense-roofline > $out/bench_quick_$r.json 2> $out/bench.err
python - <<PY
import json
j=json.loads([l for l in open('$out/bench_quick_$r.json') if l.startswith('{')][-1])
print('value', j['value'], 'chol ms', j['roofline']['ms_per_launch'], 'frac', j['roofline']['frac'], 'mem', j['device_memory_per_handle_bytes'])
PY
done
