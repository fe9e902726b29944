#!/bin/bash
# tools/r4_session_d.sh <tag> -- quick GPU check of a change of the dataflow factorisation: Cholesky parity tests, a short bench line
# (Python host, no CPU leg), a trace summary.
out=gpurun_out/${1:-r4d}; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py tests/test_gpu_dataflow_protocol.py -x -q 2>&1 | tail -8 > $out/gpu_tests.log
tail -3 $out/gpu_tests.log
timeout 300 python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $out/bench.json 2> $out/bench.err

timeout 300 python tools/df_trace.py --raw > $out/df_trace_summary.txt 2> $out/df_trace.err; cp gpurun_out/df_trace_raw.npz $out/ 2>/dev/null
python - <<PY
import json
for f in ('bench','bench_sphere2500'):
    try:
        j=json.load(open('$out/'+f+'.json')); print(f, round(j['value'],2), round(j['ms_per_step'],3), {k:round(v,3) for k,v in j['phase_ms_per_call'].items() if k in ('cholesky','schur','solve')})
    except Exception as e: print(f,'failed',e)
PY
