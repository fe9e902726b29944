#!/bin/bash
# tools/r05_final_session.sh [tag] -- the round's closing GPU session on the shipped code objects (tools/device_code_hash.sh of the same tree goes
# next to the results): GPU suite, the driver's bench command, chain trace, C++ host set-up breakdown, multi-handle stress, rocprofv3 kernel
# trace + HBM PMC passes (separate runs, single-kernel form for the counters), bench lines of the other workloads.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r05_final_session.sh r05final'
REPO=${GRAFT_REPO_ROOT:-$PWD}
out=$REPO/gpurun_out/${1:-r05final}; mkdir -p $out
cd $REPO
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $out/gpu_tests.log
tail -3 $out/gpu_tests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 300 python tools/df_trace.py > $out/df_trace_summary.txt 2> $out/df_trace.err
GTG_DEBUG_TIMING=1 timeout 600 python tools/time_sfm_bal.py ladybug1723 > $out/time_sfm_bal_cpp.json 2> $out/cpp_host_setup_breakdown.txt
timeout 300 python tools/df_stress.py 90 3 > $out/stress.txt 2> $out/stress.err
tail -1 $out/stress.txt; tail -c 300 $out/df_trace_summary.txt
for w in venice1778 dubrovnik16; do timeout 600 python bench.py --workload $w --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off > $out/bench_$w.json 2> $out/bench_$w.err; done
for w in sphere2500 w20000; do timeout 300 python bench.py --workload $w --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $out/bench_$w.json 2> $out/bench_$w.err; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w
B="python $REPO/bench.py --steps 4 --warmup 1 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- $B > $out/trace.log 2>&1
python $REPO/tools/rocprof_top.py $(find /tmp/prof_s -name "*.db" | head -1) $out/kernel_stats.csv | head -14 | cut -c1-70,190-
GTG_DF_SINGLE=1 timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python $REPO/bench.py --steps 2 --warmup 0 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $out/fetch.log 2>&1
python $REPO/tools/rocprof_pmc.py $(find /tmp/prof_f -name "*.db" | head -1) $out/pmc_fetch_size.csv > /dev/null
GTG_DF_SINGLE=1 timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python $REPO/bench.py --steps 2 --warmup 0 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $out/write.log 2>&1
python $REPO/tools/rocprof_pmc.py $(find /tmp/prof_w -name "*.db" | head -1) $out/pmc_write_size.csv > /dev/null
python $REPO/tools/pmc_traffic.py $out/pmc_fetch_size.csv $out/pmc_write_size.csv $out/pmc_cholesky_traffic.json
rm -rf /tmp/prof_s /tmp/prof_f /tmp/prof_w
cd $REPO
python - <<PY
import json
j=json.load(open('$out/bench.json'))
print('value', j['value'], 'ms', j['ms_per_step'], 'python', j['python_mirror']['value'], 'ttc cold/warm', j['time_to_converged_s'], j['time_to_converged_warm_s'])
print(j['phase_ms_per_call']); r=j['roofline']; print('frac', r['frac'], 'ms', r['ms_per_launch'], 'traffic', r['traffic'], str(r['traffic_source'])[:40])
print('lin', j['roofline_linearize']['frac'], j['roofline_linearize']['ms_per_launch'], 'mem', j.get('device_memory_per_handle_bytes'), 'cpu', j['cpu_baseline']['value'], j['cpu_baseline'].get('cores_used'))
print('parity', j.get('parity_vs_reference', {}).get('within_tolerance'), j.get('parity_vs_reference', {}).get('delta_norminf_rel'))
for w in ('venice1778','dubrovnik16','sphere2500','w20000'):
    try:
        k=json.load(open('$out/bench_%s.json' % w)); print(w, round(k['value'],2), 'it/s', round(k['lambda_tries_per_s'],2), 'tries/s', 'chol %.3f' % k['phase_ms_per_call']['cholesky'], 'mem', k.get('device_memory_per_handle_bytes'))
    except Exception as e: print(w, 'failed', e)
PY
