#!/bin/bash
# tools/next_round_first_session.sh [tag] [nostreams] -- what round 4 left for the first GPU minutes of the next round (through gpurun, ~16 GPU-min):
#   1. the grouped Schur complement's A/B against the default (bit-identity of factor / step / LM trace, four workloads), then its
#      bench lines on the L1723 and Venice shapes next to the default's (phase_ms_per_call.schur is the number);
#   2. the windowed pivot chain (libgtsam_amd_window.so): parity files, bench lines and chain traces next to the default library's;
#   3. the legacy tree stream schedule in a loop (the one test that hung once, profiles/r04_streams_tree_hang.txt): 40 runs of the
#      test's child under a 120 s bound each; a run that does not return leaves the runtime's log of its last seconds behind.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/next_round_first_session.sh r05a'
out=gpurun_out/${1:-r05a}; mkdir -p $out
GTG_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_schur_groups.py -x -q -s 2>&1 | tail -25 > $out/schur_groups_ab.log
tail -3 $out/schur_groups_ab.log
GTG_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_gpu_window_variant.py -x -q 2>&1 | tail -12 > $out/window_ab.log
tail -2 $out/window_ab.log
B="python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
for w in ladybug1723 venice1778; do
  timeout 300 $B --workload $w > $out/bench_${w}_pairs.json 2> $out/bench_${w}_pairs.err
  GTG_SCHUR=groups timeout 300 $B --workload $w > $out/bench_${w}_groups.json 2> $out/bench_${w}_groups.err
  GTG_SCHUR=groups_pipe timeout 300 $B --workload $w > $out/bench_${w}_groups_pipe.json 2> $out/bench_${w}_groups_pipe.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/bench_*_*.json')):
    try:
        j = json.load(open(f)); print(f.split('/')[-1], round(j['value'], 2), 'it/s; schur', round(j['phase_ms_per_call']['schur'], 3), 'ms; error', j['converged_error'])
    except Exception as e:
        print(f, 'failed', open(f.replace('.json', '.err')).read()[-300:])
PY
# ---- the windowed pivot chain (libgtsam_amd_window.so: chol_device.h GT_POTRF_WINDOW=1; bit-identical to the default on host threads):
# the parity files with it, then its bench line and chain trace next to the default's (phase_ms_per_call.cholesky, period_us)
W=$PWD/gtsam_amd/lib/libgtsam_amd_window.so
D=$PWD/gtsam_amd/lib/libgtsam_amd_defer.so     # the deferred last-slice update (GT_DF_DEFER_SLICE=1; its stream-schedule kernels are the product library's)
GTSAM_AMD_LIB=$W timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu 2>&1 | tail -8 > $out/window_parity.log
tail -2 $out/window_parity.log
GTSAM_AMD_LIB=$D timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu 2>&1 | tail -8 > $out/defer_parity.log
tail -2 $out/defer_parity.log
for rep in 1 2; do
  timeout 300 $B > $out/bench_default_$rep.json 2> $out/bench_default_$rep.err
  GTSAM_AMD_LIB=$W timeout 300 $B > $out/bench_window_$rep.json 2> $out/bench_window_$rep.err
  GTSAM_AMD_LIB=$D timeout 300 $B > $out/bench_defer_$rep.json 2> $out/bench_defer_$rep.err
done
timeout 300 python tools/df_trace.py > $out/df_trace_default.txt 2> /dev/null
GTSAM_AMD_LIB=$W timeout 300 python tools/df_trace.py > $out/df_trace_window.txt 2> /dev/null
for w in sphere2500 w20000; do
  timeout 300 $B --workload $w > $out/bench_${w}_default.json 2> /dev/null
  GTSAM_AMD_LIB=$W timeout 300 $B --workload $w > $out/bench_${w}_window.json 2> /dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/bench_default_*.json') + glob.glob('$out/bench_window_*.json') + glob.glob('$out/bench_defer_*.json') + glob.glob('$out/bench_sphere2500_*.json') + glob.glob('$out/bench_w20000_*.json')):
    try:
        j = json.load(open(f)); print(f.split('/')[-1], round(j['lambda_tries_per_s'], 2), 'tries/s; cholesky', round(j['phase_ms_per_call']['cholesky'], 3), 'ms; error', repr(j['converged_error']))
    except Exception as e:
        print(f, 'failed', e)
for f in ('$out/df_trace_default.txt', '$out/df_trace_window.txt'):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'period p10/p50/p90', j['period_us_p10_p50_p90'], 'mean', j['period_us_mean'])
    except Exception as e:
        print(f, 'failed', e)
PY
[ "$2" = "nostreams" ] && exit 0
# ---- the tree stream schedule, in a loop
python - > $out/nd_child.py <<PY
import re
src = open('tests/test_gpu_parity.py').read()
code = re.search(r'_ND_CHILD = r"""(.*?)"""', src, re.S).group(1)
print(code % {"root": "$PWD", "sched": "streams", "depth": 2})
PY
ok=0; hung=0
for i in $(seq 1 40); do
  # odd runs with the stream priorities the schedule had when it hung (GTG_TREE_PRIO=1), even runs without (the default since then)
  if [ $((i % 2)) = 1 ]; then export GTG_TREE_PRIO=1; else unset GTG_TREE_PRIO; fi
  if GTG_CHOL=streams GTG_ND_DEPTH=2 timeout 120 python $out/nd_child.py > $out/nd_last.out 2> $out/nd_last.err; then ok=$((ok+1));
  else
    rc=$?; hung=$((hung+1)); cp $out/nd_last.err $out/nd_failed_$i.err
    echo "run $i (GTG_TREE_PRIO=${GTG_TREE_PRIO:-0}): rc $rc" >> $out/streams_loop.txt
    # once more with the runtime's log, in case it is reproducible on this box
    GTG_CHOL=streams GTG_ND_DEPTH=2 AMD_LOG_LEVEL=3 timeout 120 python $out/nd_child.py > /dev/null 2> $out/nd_amdlog_$i.err; tail -c 20000 $out/nd_amdlog_$i.err > $out/nd_amdlog_$i.tail; rm -f $out/nd_amdlog_$i.err
    [ $hung -ge 3 ] && break
  fi
done
echo "tree stream schedule: $ok clean, $hung not clean (rc 124 = did not return within 120 s)" | tee -a $out/streams_loop.txt
