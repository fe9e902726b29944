#!/bin/bash
out=gpurun_out/r4f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_device_analysis.py tests/test_gpu_speculative.py -x -q 2>&1 | tail -15 > $out/gpu_tests.log; tail -4 $out/gpu_tests.log
GTG_DEBUG_TIMING=1 timeout 600 python tools/time_sfm_bal.py ladybug1723 > $out/time_sfm_bal_cpp.json 2> $out/cpp_host_setup_breakdown.txt
grep -n "ordering\|tile schedule\|library:" $out/cpp_host_setup_breakdown.txt | tail -8
