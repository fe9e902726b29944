#!/bin/bash
# tools/gpu_session.sh <tag> -- one GPU session through gpurun: the GPU test suite, the default bench line, the shim's test driver and
# the reference's timeSFMBAL program through the C++ host.  Writes gpurun_out/<tag>/ (copy what is to be kept into profiles/).
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/gpu_session.sh r04a'
out=gpurun_out/${1:-session}; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $out/gpu_tests.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
timeout 600 tests/_build/test_gpu_lm_gtsam > $out/shim_test.log 2>&1
GTG_DEBUG_TIMING=1 timeout 600 python tools/time_sfm_bal.py ladybug1723 > $out/time_sfm_bal_cpp.json 2> $out/cpp_host_setup_breakdown.txt
tail -3 $out/gpu_tests.log; tail -c 400 $out/bench.json; tail -2 $out/shim_test.log
