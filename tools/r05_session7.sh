#!/bin/bash
# tools/r05_session7.sh [tag] -- sweep of the diagonal tiles' last piece (GT_DF_FINAL_DIAG = 1 / 2 (default) / 3 / 4 (rounds 2-4)): bench lines interleaved, chain trace
out=gpurun_out/${1:-r05g}; mkdir -p $out
B="python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
L=$PWD/gtsam_amd/lib
for rep in 1 2 3; do
  for v in default nr3; do
    if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so; fi
    timeout 200 $B > $out/ab_${v}_$rep.json 2> $out/ab_${v}_$rep.err
  done
done
for v in default; do
  if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so; fi
  timeout 200 python tools/df_trace.py --raw > $out/df_trace_$v.txt 2> $out/df_trace_$v.err; cp gpurun_out/df_trace_raw.npz $out/df_trace_raw_$v.npz 2>/dev/null
  timeout 300 $B --workload venice1778 > $out/ab_${v}_venice1778.json 2> $out/ab_${v}_venice1778.err
done
unset GTSAM_AMD_LIB
timeout 300 python -m pytest tests/test_gpu_headline_parity.py -x -q -m gpu 2>&1 | tail -2
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
        print(f, 'failed', e)
PY
