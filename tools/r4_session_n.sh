#!/bin/bash
out=gpurun_out/r4n; mkdir -p $out
for w in 0 2 3 5; do
  GTG_DF_PULL=$w timeout 300 python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $out/bench_pull$w.json 2> $out/bench_pull$w.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_pull*.json')):
    j=json.load(open(f)); print(f.split('/')[-1], round(j['value'],2), 'chol', round(j['phase_ms_per_call']['cholesky'],3), 'err', j['converged_error'])
PY
