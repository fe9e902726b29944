#!/bin/bash
# tools/r06_shim_check.sh [tag] -- the C++ shim's GPU tests, then five runs of the C++ bench program: warm construction / time-to-converged
REPO=${GRAFT_REPO_ROOT:-$PWD}
out=$REPO/gpurun_out/${1:-r14a}; mkdir -p $out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_gtsam_shim.py tests/test_gpu_headline_parity.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
python - <<PY
import sys; sys.path.insert(0, '.')
import bench
bench.write_workload_file("ladybug1723", "/tmp/l1723.txt")
PY
for i in 1 2 3 4 5; do
  GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 7 --warmup 0 > $out/run_$i.json 2> $out/run_$i.txt
  python - <<PY
import json
j=json.loads([l for l in open('$out/run_$i.json') if l.startswith('{')][-1])
print('run $i', {k: j[k] for k in ('cold_construct_ms','cold_optimize_ms','cold_time_to_converged_s','warm_construct_ms','warm_optimize_ms','warm_time_to_converged_s')})
PY
  awk '/threads started/{n++} n==2' $out/run_$i.txt | grep -E "walk|extraction|merge|library:|wait for the copies" | awk '{print $(NF-1)}' | tr '\n' ' '; echo
done
