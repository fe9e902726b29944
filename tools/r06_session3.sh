#!/bin/bash
# tools/r06_session3.sh [tag] -- cold / warm time-to-converged of the C++ host after the prewarm thread stopped blocking gtg_create
out=gpurun_out/${1:-r06d}; mkdir -p $out
python - <<PY
import sys; sys.path.insert(0, '.')
import bench
bench.write_workload_file("ladybug1723", "/tmp/l1723.txt")
PY
for rep in 1 2 3; do GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 20 --warmup 5 > $out/cpp_bench_$rep.json 2> $out/cpp_setup_breakdown_$rep.txt; done
for rep in 4 5; do tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 20 --warmup 5 > $out/cpp_bench_$rep.json 2> /dev/null; done
python - <<PY
import json
for rep in (1, 2, 3, 4, 5):
    j = json.loads([l for l in open('$out/cpp_bench_%d.json' % rep) if l.startswith('{')][-1])
    print(rep, {k: j[k] for k in ('iterations_per_s', 'cold_construct_ms', 'cold_optimize_ms', 'cold_time_to_converged_s', 'warm_construct_ms', 'warm_optimize_ms', 'warm_time_to_converged_s', 'final_error')})
PY
grep "shim " $out/cpp_setup_breakdown_1.txt | head -32
timeout 600 python -m pytest tests/test_gpu_gtsam_shim.py -x -q 2>&1 | tail -3
