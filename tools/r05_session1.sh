#!/bin/bash
# tools/r05_session1.sh [tag] -- first GPU session of round 5: measure what round 4 built without a GPU.
#   1. grouped Schur complement: bit-identity A/B (tests/test_gpu_schur_groups.py), bench lines pairs / groups / groups_pipe on L1723 + Venice
#   2. chain variants: windowed pivot chain (_window), deferred last slice (_defer), RMW-first flag polls (_safe2): parity, bench, chain trace
out=gpurun_out/${1:-r05a}; mkdir -p $out
GTG_TEST_EXPERIMENTAL=1 timeout 400 python -m pytest tests/test_gpu_schur_groups.py -x -q -s 2>&1 | tail -25 > $out/schur_groups_ab.log
tail -3 $out/schur_groups_ab.log
GTG_TEST_EXPERIMENTAL=1 timeout 400 python -m pytest tests/test_gpu_window_variant.py -x -q 2>&1 | tail -12 > $out/window_ab.log
tail -2 $out/window_ab.log
B="python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
for w in ladybug1723 venice1778; do
  timeout 200 $B --workload $w > $out/bench_${w}_pairs.json 2> $out/bench_${w}_pairs.err
  GTG_SCHUR=groups timeout 200 $B --workload $w > $out/bench_${w}_groups.json 2> $out/bench_${w}_groups.err
  GTG_SCHUR=groups_pipe timeout 200 $B --workload $w > $out/bench_${w}_groups_pipe.json 2> $out/bench_${w}_groups_pipe.err
done
GTG_SCHUR=groups GTG_SCHUR_LISTS=device timeout 200 $B --workload ladybug1723 > $out/bench_ladybug1723_groups_devlists.json 2> $out/bench_ladybug1723_groups_devlists.err
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/bench_*_*.json')):
    try:
        j = json.load(open(f)); print(f.split('/')[-1], round(j['value'], 2), 'it/s; schur', round(j['phase_ms_per_call']['schur'], 3), 'ms; error', j['converged_error'], 'setup', j.get('time_to_converged_setup_s'))
    except Exception as e:
        print(f, 'failed', open(f.replace('.json', '.err')).read()[-300:])
PY
L=$PWD/gtsam_amd/lib
GTSAM_AMD_LIB=$L/libgtsam_amd_window.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu 2>&1 | tail -8 > $out/window_parity.log
tail -2 $out/window_parity.log
for rep in 1 2; do
  for v in default window defer safe2; do
    if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so; fi
    timeout 200 $B > $out/chain_${v}_$rep.json 2> $out/chain_${v}_$rep.err
  done
done
for v in default window defer safe2; do
  if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so; fi
  timeout 200 python tools/df_trace.py > $out/df_trace_$v.txt 2> /dev/null
  for w in sphere2500; do timeout 200 $B --workload $w > $out/chain_${v}_$w.json 2> /dev/null; done
done
unset GTSAM_AMD_LIB
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/chain_*.json')):
    try:
        j = json.load(open(f)); print(f.split('/')[-1], round(j['lambda_tries_per_s'], 2), 'tries/s; cholesky', round(j['phase_ms_per_call']['cholesky'], 3), 'ms; error', repr(j['converged_error']))
    except Exception as e:
        print(f, 'failed', e)
for f in sorted(glob.glob('$out/df_trace_*.txt')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'period p10/p50/p90', j['period_us_p10_p50_p90'], 'mean', j['period_us_mean'], {k: v for k, v in j.items() if 'rmw' in k or 'shadow' in k})
    except Exception as e:
        print(f, 'failed', e)
PY
