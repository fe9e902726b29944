#!/bin/bash
out=gpurun_out/r4j; mkdir -p $out
for w in w20000 sphere2500; do
  for cfg in "8 4" "8 5" "16 4" "16 5"; do
    set -- $cfg
    GTG_DF_SLOTS=$1 GTG_ND_DEPTH=$2 timeout 300 python bench.py --workload $w --steps 12 --warmup 3 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $out/bench_${w}_s$1_d$2.json 2> $out/bench_${w}_s$1_d$2.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        j=json.load(open(f)); print(f.split('/')[-1], round(j['value'],1), 'chol', round(j['phase_ms_per_call']['cholesky'],3), 'ms/step', round(j['ms_per_step'],3), 'err', j['converged_error'])
    except Exception as e: print(f,'failed',e, open(f.replace('.json','.err')).read()[-300:])
PY
