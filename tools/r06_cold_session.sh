#!/bin/bash
# tools/r06_cold_session.sh [tag] -- where the first optimize() of a cold process spends its time (per-iteration wall times, three cold processes)
REPO=${GRAFT_REPO_ROOT:-$PWD}
out=$REPO/gpurun_out/${1:-r09g}; mkdir -p $out
cd $REPO
python - <<PY
import sys; sys.path.insert(0, '.')
import bench
bench.write_workload_file("ladybug1723", "/tmp/l1723.txt")
PY
for i in 1 2 3; do
  GTG_DEBUG_TIMING=1 tests/_build/bench_lm_gtsam /tmp/l1723.txt --steps 7 --warmup 0 > $out/cold_$i.json 2> $out/cold_$i.txt
  echo "--- process $i"; grep -E "first \]|iteration 1|optimize\(\)|library:" $out/cold_$i.txt | head -22 | cut -c1-100
done
