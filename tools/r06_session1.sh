#!/bin/bash
# tools/r06_session1.sh [tag] -- first GPU session of round 6: the C++ shim test (GpuState identity, back-to-back / rejected base-class tryLambda),
# the two C++ bench programs through bench.py's new `workloads` legs, with the wall-clock of the whole default line.
out=gpurun_out/${1:-r06a}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_gtsam_shim.py -x -q 2>&1 | tail -25 > $out/shim_tests.log; tail -4 $out/shim_tests.log
t0=$(date +%s)
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench wall $(( $(date +%s) - t0 )) s, rc $?" | tee $out/bench_wall.txt
tail -c 600 $out/bench.err
python - <<PY
import json
j=json.loads([l for l in open('$out/bench.json') if l.startswith('{')][-1])
print('value', j['value'], 'ms', j['ms_per_step'], 'ttc cold/warm', j['time_to_converged_s'], j['time_to_converged_warm_s'])
print('cpu', {k: j['cpu_baseline'].get(k) for k in ('value','cores','kind')}, 'assisted', (j['cpu_baseline'].get('assisted') or {}).get('value'))
for w, r in (j.get('workloads') or {}).items():
    print(w, {k: r.get(k) for k in ('value','lambda_tries_per_s','time_to_converged_cold_s','time_to_converged_warm_s','converged_error','converged_iterations','converged_inner_iterations','failed','trajectory_matches_reference')})
    print('   roofline', {k: (r.get('roofline') or {}).get(k) for k in ('frac','ms_per_launch','launches','frac_stored_tiles')}, 'cpu', {k: (r.get('cpu_baseline') or {}).get(k) for k in ('value','optimize_ms','iterations','inner_iterations','final_error','error_trace_max_rel_diff_vs_device')})
    print('   phases/try', r.get('device_phase_ms_per_try'))
PY
