#!/bin/bash
# tools/r06_slots_sweep.sh [tag] -- chain slots of the dataflow factorisation on the pose graphs (GTG_DF_SLOTS; 8 = what ships)
out=gpurun_out/${1:-r06j}; mkdir -p $out
for w in sphere2500 w20000; do for sl in 8 12 16; do for nd in auto 5; do
if [ $nd = auto ]; then unset GTG_ND_DEPTH; else export GTG_ND_DEPTH=$nd; fi
GTG_DF_SLOTS=$sl timeout 600 python bench.py --workload $w --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $out/b.json 2> $out/b.err
python - <<PY
import json
j=json.loads([l for l in open('$out/b.json') if l.startswith('{')][-1])
print('$w', 'slots', $sl, 'nd', '$nd', 'tries/s', round(j['lambda_tries_per_s'],1), 'chol ms', round(j['phase_ms_per_call']['cholesky'],4), 'err', j['converged_error'], 'its', j['converged_iterations'], j['converged_inner_iterations'])
PY
done; done; done | tee $out/slots_sweep.txt
