#!/bin/bash
# tools/r05_session8.sh [tag] -- the whole GPU suite on the current sources, then bench lines (Python mirror) of every workload
out=gpurun_out/${1:-r05k}; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $out/gpu_tests.log; tail -3 $out/gpu_tests.log
B="python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
for rep in 1 2; do timeout 200 $B > $out/ab_default_$rep.json 2> $out/ab_default_$rep.err; done
for w in venice1778 sphere2500 w20000 dubrovnik16; do timeout 300 $B --workload $w > $out/ab_default_$w.json 2> $out/ab_default_$w.err; done
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/ab_*.json')):
    try:
        j = json.load(open(f)); ph = j['phase_ms_per_call']
        print(f.split('/')[-1], round(j['value'], 2), 'it/s', round(j['lambda_tries_per_s'], 2), 'tries/s;', ' '.join('%s %.3f' % (k, v) for k, v in ph.items()), '; error', repr(j['converged_error']))
    except Exception as e:
        print(f, 'failed', e, open(f.replace('.json', '.err')).read()[-400:])
PY
