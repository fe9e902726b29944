#!/bin/bash
# tools/r05_session3.sh [tag] -- round 5, third GPU session: the hand-over between consecutive diagonal tiles (chain_loop) on hardware.
#   1. parity + dataflow protocol (multi-handle bit-identity, fenced A/B, forced time-out) + C++ shim tests
#   2. A/B: default / _nohandover (GT_DF_HANDOVER=0) / _records (stored Jacobian records): bench lines, chain traces
#   3. kernel stats of the default build
out=gpurun_out/${1:-r05c}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py tests/test_gpu_dataflow_protocol.py tests/test_gpu_gtsam_shim.py -x -q -m gpu 2>&1 | tail -40 > $out/tests.log
tail -4 $out/tests.log
B="python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
L=$PWD/gtsam_amd/lib
for rep in 1 2; do
  for v in default nohandover records; do
    if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so; fi
    timeout 200 $B > $out/ab_${v}_$rep.json 2> $out/ab_${v}_$rep.err
  done
done
for v in default nohandover; do
  if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so; fi
  timeout 200 python tools/df_trace.py --raw > $out/df_trace_$v.txt 2> $out/df_trace_$v.err; cp gpurun_out/df_trace_raw.npz $out/df_trace_raw_$v.npz 2>/dev/null
  for w in venice1778 sphere2500 w20000 dubrovnik16; do timeout 300 $B --workload $w > $out/ab_${v}_$w.json 2> $out/ab_${v}_$w.err; done
done
unset GTSAM_AMD_LIB
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/ab_*.json')):
    try:
        j = json.load(open(f)); ph = j['phase_ms_per_call']
        print(f.split('/')[-1], round(j['value'], 2), 'it/s', round(j['lambda_tries_per_s'], 2), 'tries/s;', ' '.join('%s %.3f' % (k, v) for k, v in ph.items()), '; error', repr(j['converged_error']))
    except Exception as e:
        print(f, 'failed', e, open(f.replace('.json', '.err')).read()[-400:])
for f in sorted(glob.glob('$out/df_trace_*.txt')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'total', j['total_us'], 'period p10/p50/p90', [round(x, 2) for x in j['period_us_p10_p50_p90']], 'mean', round(j['period_us_mean'], 2), 'sub_after_potrf', round(j['sub_after_potrf_us_mean'], 2))
    except Exception as e:
        print(f, 'failed', e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r05c -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $GRAFT_REPO_ROOT/$out/prof_bench.json 2> $GRAFT_REPO_ROOT/$out/prof_bench.err
cd $GRAFT_REPO_ROOT
python tools/rocprof_top.py $(find /tmp/prof_r05c -name "*.db" | head -1) $out/kernel_stats.csv > /dev/null 2> $out/kernel_stats.err
head -24 $out/kernel_stats.csv | cut -c1-60,200-
