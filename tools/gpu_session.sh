#!/bin/bash
# tools/gpu_session.sh <tag> -- one GPU session through gpurun: the GPU test suite, the default bench line (C++ host headline, in-run PMC
# traffic, reference CPU leg), the dataflow trace, the C++ host's set-up breakdown, a multi-handle stress.  Writes gpurun_out/<tag>/
# (copy what is to be kept into profiles/).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_session.sh r04final'
out=gpurun_out/${1:-session}; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $out/gpu_tests.log
tail -3 $out/gpu_tests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 300 python tools/df_trace.py > $out/df_trace_summary.txt 2> $out/df_trace.err
GTG_DEBUG_TIMING=1 timeout 600 python tools/time_sfm_bal.py ladybug1723 > $out/time_sfm_bal_cpp.json 2> $out/cpp_host_setup_breakdown.txt
timeout 300 python tools/df_stress.py 120 3 > $out/stress.txt 2> $out/stress.err
tail -1 $out/stress.txt; tail -c 300 $out/df_trace_summary.txt
python - <<PY
import json
j=json.load(open('$out/bench.json'))
print('value', j['value'], 'ms', j['ms_per_step'], 'python', j['python_mirror']['value'], 'ttc cold/warm', j['time_to_converged_s'], j['time_to_converged_warm_s'])
print(j['phase_ms_per_call']); r=j['roofline']; print('frac', r['frac'], 'ms', r['ms_per_launch'], 'traffic', r['traffic'], r['traffic_source'][:40])
print('mem', j.get('device_memory_per_handle_bytes'), 'cpu', j['cpu_baseline']['value'], j['cpu_baseline'].get('cores_used'))
PY
