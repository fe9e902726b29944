#!/bin/bash
# tools/r4_session_e.sh -- round 4: speculative lambda search (replicas on one GPU, gloo), set-up breakdown through the C++ host, the
# round's profile (bench line, rocprofv3 kernel stats, PMC passes).
out=gpurun_out/r4e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_speculative.py -x -q 2>&1 | tail -12 > $out/gpu_speculative.log; tail -3 $out/gpu_speculative.log
GTG_DEBUG_TIMING=1 timeout 600 python tools/time_sfm_bal.py ladybug1723 > $out/time_sfm_bal_cpp.json 2> $out/cpp_host_setup_breakdown.txt
bash tools/profile_round.sh r04
mkdir -p $out/prof; mv gpurun_out/r04_* $out/prof/ 2>/dev/null
tail -30 $out/cpp_host_setup_breakdown.txt
