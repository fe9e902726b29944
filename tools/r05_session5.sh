#!/bin/bash
# tools/r05_session5.sh [tag] -- hand-over iterations: default build only (the no-hand-over numbers of the same sources are in r05d), bench x2 + raw trace
out=gpurun_out/${1:-r05e}; mkdir -p $out
B="python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
timeout 300 python -m pytest tests/test_gpu_headline_parity.py -x -q -m gpu -k "ladybug1723" 2>&1 | tail -3 > $out/tests.log; tail -1 $out/tests.log
for rep in 1 2; do timeout 200 $B > $out/ab_default_$rep.json 2> $out/ab_default_$rep.err; done
timeout 200 python tools/df_trace.py --raw > $out/df_trace_default.txt 2> $out/df_trace_default.err; cp gpurun_out/df_trace_raw.npz $out/df_trace_raw_default.npz 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/ab_*.json')):
    try:
        j = json.load(open(f)); ph = j['phase_ms_per_call']
        print(f.split('/')[-1], round(j['value'], 2), 'it/s', round(j['lambda_tries_per_s'], 2), 'tries/s; cholesky %.3f' % ph['cholesky'], '; error', repr(j['converged_error']))
    except Exception as e:
        print(f, 'failed', e, open(f.replace('.json', '.err')).read()[-400:])
for f in sorted(glob.glob('$out/df_trace_*.txt')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'total', j['total_us'], 'period p10/p50/p90', [round(x, 2) for x in j['period_us_p10_p50_p90']], 'mean', round(j['period_us_mean'], 2))
    except Exception as e:
        print(f, 'failed', e, open(f.replace('.txt','.err')).read()[-300:])
PY
