#!/bin/bash
# tools/r06_session2.sh [tag] -- hand-written primitives in the place of rocPRIM, gtg_prewarm, the restructured shim constructor
out=gpurun_out/${1:-r06c}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_device_analysis.py -x -q 2>&1 | tail -15 > $out/primitives_tests.log; tail -3 $out/primitives_tests.log
python - <<PY
import sys; sys.path.insert(0, '.')
import bench
bench.write_workload_file("ladybug1723", "/tmp/l1723.txt")
PY
for rep in 1 2; do tests/_build/cold_start_probe /tmp/l1723.txt > $out/cold_probe_$rep.txt 2>&1; tests/_build/cold_start_probe /tmp/l1723.txt --prewarm > $out/cold_probe_prewarm_$rep.txt 2>&1; done
cat $out/cold_probe_1.txt; echo; cat $out/cold_probe_prewarm_1.txt
for rep in 1 2 3; do GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 20 --warmup 5 > $out/cpp_bench_$rep.json 2> $out/cpp_setup_breakdown_$rep.txt; done
python - <<PY
import json
for rep in (1, 2, 3):
    j = json.loads([l for l in open('$out/cpp_bench_%d.json' % rep) if l.startswith('{')][-1])
    print(rep, {k: j[k] for k in ('iterations_per_s', 'cold_construct_ms', 'cold_optimize_ms', 'cold_time_to_converged_s', 'warm_construct_ms', 'warm_optimize_ms', 'warm_time_to_converged_s', 'final_error')})
PY
grep "shim \|setup\]" $out/cpp_setup_breakdown_1.txt | head -60
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $out/gpu_tests.log; tail -3 $out/gpu_tests.log
