#!/bin/bash
# tools/r05_session2.sh [tag] -- round 5, second GPU session: the pruned library (no tree stream schedule, deferred last slice, no variant
# switches) and the fused linearisation (fused.h) on hardware: full GPU suite, A/B against the stored-record build, kernel times, chain trace.
out=gpurun_out/${1:-r05b}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu 2>&1 | tail -15 > $out/parity.log
tail -3 $out/parity.log
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_headline_parity.py 2>&1 | tail -15 > $out/suite_rest.log
tail -3 $out/suite_rest.log
B="python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
L=$PWD/gtsam_amd/lib
for rep in 1 2; do
  for v in default records; do
    if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so; fi
    timeout 200 $B > $out/lin_${v}_$rep.json 2> $out/lin_${v}_$rep.err
  done
done
for v in default records; do
  if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so; fi
  timeout 300 $B --workload venice1778 > $out/lin_${v}_venice.json 2> $out/lin_${v}_venice.err
  timeout 200 $B --workload dubrovnik16 > $out/lin_${v}_dubrovnik16.json 2> $out/lin_${v}_dubrovnik16.err
done
unset GTSAM_AMD_LIB
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/lin_*.json')):
    try:
        j = json.load(open(f)); ph = j['phase_ms_per_call']
        print(f.split('/')[-1], round(j['value'], 2), 'it/s', round(j['lambda_tries_per_s'], 2), 'tries/s;', ' '.join('%s %.3f' % (k, v) for k, v in ph.items()), '; error', repr(j['converged_error']), 'mem', j.get('device_memory_per_handle_bytes'))
    except Exception as e:
        print(f, 'failed', e, open(f.replace('.json', '.err')).read()[-400:])
PY
# per-kernel times of the default library (kernel trace does not serialise: the production form)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r05b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $GRAFT_REPO_ROOT/$out/prof_bench.json 2> $GRAFT_REPO_ROOT/$out/prof_bench.err
cd $GRAFT_REPO_ROOT
python tools/rocprof_top.py $(find /tmp/prof_r05b -name "*.db" | head -1) $out/kernel_stats.csv > /dev/null 2> $out/kernel_stats.err
head -30 $out/kernel_stats.csv | cut -c1-150
timeout 200 python tools/df_trace.py --raw > $out/df_trace.txt 2> $out/df_trace.err; cp gpurun_out/df_trace_raw.npz $out/ 2>/dev/null
tail -1 $out/df_trace.txt | cut -c1-400
