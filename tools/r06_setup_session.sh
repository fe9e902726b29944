#!/bin/bash
# tools/r06_setup_session.sh [tag] -- set-up passes: device-vs-host parity of the symbolic passes + the C++ host's set-up breakdown on L1723
REPO=${GRAFT_REPO_ROOT:-$PWD}
out=$REPO/gpurun_out/${1:-r09a}; mkdir -p $out
cd $REPO
timeout 900 python -m pytest tests/test_gpu_device_analysis.py tests/test_gpu_gtsam_shim.py -x -q 2>&1 | tail -15 > $out/tests.txt
tail -3 $out/tests.txt
python - <<PY
import sys; sys.path.insert(0, '.')
import bench
bench.write_workload_file("ladybug1723", "/tmp/l1723.txt")
PY
GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 20 --warmup 5 > $out/cpp_bench.json 2> $out/cpp_host_setup_breakdown.txt
GTG_HOST_SYMBOLIC=1 GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 20 --warmup 5 > $out/cpp_bench_hostsym.json 2> $out/cpp_host_setup_breakdown_hostsym.txt
awk "/threads started/{n++} n==2" $out/cpp_host_setup_breakdown.txt | cut -c1-120
echo ---- host symbolic
grep -E "tile structure|resolved|library:" $out/cpp_host_setup_breakdown_hostsym.txt | tail -4
python - <<PY
import json
for f in ('cpp_bench.json', 'cpp_bench_hostsym.json'):
    j = json.loads([l for l in open('$out/' + f) if l.startswith('{')][-1])
    print(f, {k: j.get(k) for k in ('iterations_per_s', 'warm_construct_ms', 'time_to_converged_warm_s', 'time_to_converged_cold_s')})
PY
for th in 4 8 16; do
  GTG_HOST_THREADS=$th GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 20 --warmup 5 > $out/cpp_bench_t$th.json 2> $out/breakdown_t$th.txt
  echo "threads $th: $(awk '/threads started/{n++} n==2' $out/breakdown_t$th.txt | grep -E 'extraction|merge|library:|wait for the copies' | awk '{print $(NF-1)}' | tr '\n' ' ') $(python -c "
import json; j=json.loads([l for l in open('$out/cpp_bench_t$th.json') if l.startswith('{')][-1]); print(j['warm_construct_ms'], j['cold_construct_ms'])")"
done
