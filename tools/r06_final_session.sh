#!/bin/bash
# tools/r06_final_session.sh [tag] -- the round's closing GPU session on the shipped code objects (tools/device_code_hash.sh of the same tree goes
# next to the results): GPU suite + smoke, the driver's bench command (one line: L1723 headline + sphere2500 / Venice workloads), chain trace,
# C++ host set-up breakdown, cold-start anatomy, multi-handle stress with the poll statistics, rocprofv3 kernel trace + HBM PMC passes
# (separate runs, single-kernel form for the counters), bench lines of the remaining workloads.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/r06_final_session.sh r06final'
REPO=${GRAFT_REPO_ROOT:-$PWD}
out=$REPO/gpurun_out/${1:-r06final}; mkdir -p $out
cd $REPO
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $out/gpu_tests.log
grep -E "passed|failed" $out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_command.json 2> $out/bench.err
echo "driver command wall $(( $(date +%s) - t0 )) s" | tee $out/bench_wall.txt
timeout 900 python bench.py --workloads off > $out/bench_default_flags.json 2>> $out/bench.err
timeout 300 python tools/df_trace.py > $out/df_trace_summary.txt 2> $out/df_trace.err
python - <<PY
import sys; sys.path.insert(0, '.')
import bench
bench.write_workload_file("ladybug1723", "/tmp/l1723.txt")
PY
GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 20 --warmup 5 > $out/cpp_bench_with_breakdown.json 2> $out/cpp_host_setup_breakdown.txt
for i in 1 2 3; do GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 7 --warmup 0 > $out/cold_process_$i.json 2> $out/cold_process_$i.txt; done   # three cold processes: first-iteration anatomy
GTG_HOST_SYMBOLIC=1 GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 7 --warmup 0 > /dev/null 2> $out/cpp_host_setup_breakdown_host_symbolic.txt   # the host loops the device passes replace
tests/_build/cold_start_probe /tmp/l1723.txt > $out/cold_start_probe.txt 2>&1
tests/_build/cold_start_probe /tmp/l1723.txt --prewarm > $out/cold_start_probe_prewarmed.txt 2>&1
timeout 300 python tools/df_stress.py 90 3 > $out/stress.txt 2> $out/stress.err
tail -1 $out/stress.txt; tail -c 300 $out/df_trace_summary.txt
for w in dubrovnik16; do timeout 600 python bench.py --workload $w --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off > $out/bench_$w.json 2> $out/bench_$w.err; done
for w in w20000; do timeout 600 python bench.py --workload $w --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off > $out/bench_$w.json 2> $out/bench_$w.err; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w
B="python $REPO/bench.py --steps 4 --warmup 1 --cpu-baseline off --skip-dense-roofline --traffic off --host python --workloads off"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- $B > $out/trace.log 2>&1
python $REPO/tools/rocprof_top.py $(find /tmp/prof_s -name "*.db" | head -1) $out/kernel_stats.csv | head -16 | cut -c1-70,190-
GTG_DF_SINGLE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python $REPO/bench.py --steps 2 --warmup 0 --cpu-baseline off --skip-dense-roofline --traffic off --host python --workloads off > $out/fetch.log 2>&1
python $REPO/tools/rocprof_pmc.py $(find /tmp/prof_f -name "*.db" | head -1) $out/pmc_fetch_size.csv > /dev/null
GTG_DF_SINGLE=1 timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python $REPO/bench.py --steps 2 --warmup 0 --cpu-baseline off --skip-dense-roofline --traffic off --host python --workloads off > $out/write.log 2>&1
python $REPO/tools/rocprof_pmc.py $(find /tmp/prof_w -name "*.db" | head -1) $out/pmc_write_size.csv > /dev/null
python $REPO/tools/pmc_traffic.py $out/pmc_fetch_size.csv $out/pmc_write_size.csv $out/pmc_cholesky_traffic.json
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w
cd $REPO
python - <<PY
import json
j=json.loads([l for l in open('$out/bench_driver_command.json') if l.startswith('{')][-1])
print('value', j['value'], 'ms', j['ms_per_step'], 'python', j['python_mirror']['value'], 'ttc cold/warm', j['time_to_converged_s'], j['time_to_converged_warm_s'])
print(j['phase_ms_per_call']); r=j['roofline']; print('frac', r['frac'], 'ms', r['ms_per_launch'], 'traffic', r['traffic'], str(r['traffic_source'])[:40])
print('lin', j['roofline_linearize']['frac'], j['roofline_linearize']['ms_per_launch'], 'mem', j.get('device_memory_per_handle_bytes'), j.get('device_memory_cached_scratch_bytes'), 'cpu', j['cpu_baseline']['value'], (j['cpu_baseline'].get('assisted') or {}).get('value'))
print('parity', j.get('parity_vs_reference', {}).get('within_tolerance'), j.get('parity_vs_reference', {}).get('delta_norminf_rel'))
for w, r in (j.get('workloads') or {}).items():
    print(w, {k: r.get(k) for k in ('value','lambda_tries_per_s','time_to_converged_cold_s','time_to_converged_warm_s','converged_error','failed','trajectory_matches_reference')}, 'frac', (r.get('roofline') or {}).get('frac'), 'cpu', (r.get('cpu_baseline') or {}).get('value'))
for w in ('dubrovnik16','w20000'):
    try:
        k=json.loads([l for l in open('$out/bench_%s.json' % w) if l.startswith('{')][-1]); print(w, round(k['value'],2), 'it/s', round(k['lambda_tries_per_s'],2), 'tries/s', 'chol %.3f' % k['phase_ms_per_call']['cholesky'], 'source', k['value_source'][:20])
    except Exception as e: print(w, 'failed', e)
PY
