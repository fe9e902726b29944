#!/bin/bash
# tools/r4_session_b.sh -- round 4, second GPU session: flags published / polled by RMW atomics + same-schedule repeat, under stress;
# GPU suite; the bench line with the C++ host as the headline and in-run PMC traffic; a raw dataflow trace.
out=gpurun_out/r4b; mkdir -p $out
timeout 500 python tools/df_stress.py 240 3 > $out/stress_rmw.txt 2> $out/stress_rmw.err
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $out/gpu_tests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 300 python tools/df_trace.py --raw > $out/df_trace_summary.txt 2> $out/df_trace.err
cp gpurun_out/df_trace_raw.npz $out/ 2>/dev/null
tail -2 $out/stress_rmw.txt; tail -3 $out/gpu_tests.log; tail -c 1500 $out/bench.json; tail -c 600 $out/df_trace_summary.txt
