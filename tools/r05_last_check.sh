#!/bin/bash
# tools/r05_last_check.sh [tag] -- the GPU suite and smoke() on the final tree (host-side changes after the closing session; device code unchanged)
out=gpurun_out/${1:-r05z}; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > $out/gpu_tests.log; grep -E 'passed|failed|error' $out/gpu_tests.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
