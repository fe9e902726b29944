#!/bin/bash
# tools/r06_level_sweep.sh [tag] [levels...] -- Cholesky ms on the L1723 shape against GTG_DF_LEVEL (load levelling of the early pieces; 0 = off)
out=gpurun_out/${1:-r06i}; mkdir -p $out; shift
for rep in 1 2; do for lv in ${@:-0 100 120 150 90}; do
GTG_DF_LEVEL=$lv timeout 600 python bench.py --workloads off --cpu-baseline off --traffic off --skip-dense-roofline --host python > $out/bench_level_${lv}_$rep.json 2> $out/bench.err
python - <<PY
import json
j=json.loads([l for l in open('$out/bench_level_${lv}_$rep.json') if l.startswith('{')][-1])
print('level', $lv, 'rep', $rep, 'value', round(j['value'],2), 'chol ms', round(j['roofline']['ms_per_launch'],4), 'err', j['converged_error'])
PY
done; done | tee $out/level_sweep.txt
