#!/bin/bash
# tools/r06_cold_probe.sh [tag] -- cold-start anatomy (tools/cpp/cold_start_probe.cpp) on the L1723 shape, default and with eager code-object loading
out=gpurun_out/${1:-r06b}; mkdir -p $out
python - <<PY
import sys; sys.path.insert(0, '.')
import bench
bench.write_workload_file("ladybug1723", "/tmp/l1723.txt")
PY
for rep in 1 2; do tests/_build/cold_start_probe /tmp/l1723.txt > $out/cold_probe_$rep.txt 2>&1; done
HIP_ENABLE_DEFERRED_LOADING=0 tests/_build/cold_start_probe /tmp/l1723.txt > $out/cold_probe_eager.txt 2>&1
GTG_DEBUG_TIMING=1 tests/_build/cold_start_probe /tmp/l1723.txt > $out/cold_probe_timing.txt 2>&1
cat $out/cold_probe_1.txt; echo; cat $out/cold_probe_eager.txt; echo; tail -50 $out/cold_probe_timing.txt
