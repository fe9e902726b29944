#!/bin/bash
# tools/r4_session_a.sh -- round 4, first GPU session: the hand-off fix of the dataflow Cholesky under multi-handle stress, A/B against
# the round-3 protocol on the SAME box, then the GPU test suite and the bench line.  Writes gpurun_out/r4a/.
out=gpurun_out/r4a; mkdir -p $out
L=$PWD/gtsam_amd/lib
{ cat /opt/rocm/.info/version; rocm-smi --showdriverversion 2>/dev/null | grep -i version; nproc; } > $out/env.txt 2>&1
GTSAM_AMD_LIB=$L/libgtsam_amd_r3proto.so timeout 300 python tools/df_stress.py 90 3 > $out/stress_r3proto.txt 2> $out/stress_r3proto.err
timeout 500 python tools/df_stress.py 240 3 > $out/stress_fixed.txt 2> $out/stress_fixed.err
GTSAM_AMD_LIB=$L/libgtsam_amd_safe4.so timeout 300 python tools/df_stress.py 60 3 > $out/stress_safe4.txt 2> $out/stress_safe4.err
GTSAM_AMD_LIB=$L/libgtsam_amd_safe1.so timeout 300 python tools/df_stress.py 60 3 > $out/stress_safe1.txt 2> $out/stress_safe1.err
timeout 300 python tools/df_stress.py 60 3 bal300 > $out/stress_fixed_bal300.txt 2> $out/stress_fixed_bal300.err
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $out/gpu_tests.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
for f in $out/stress_*.txt; do echo "== $f"; tail -2 $f; done; tail -3 $out/gpu_tests.log; tail -c 600 $out/bench.json
