#!/bin/bash
# tools/r05_session9.sh [tag] -- the express queue of the dataflow Cholesky (chol_dataflow.hip::bulk_loop): parity files first, then interleaved
# bench lines (Python mirror) of the default (16 express workgroups) against builds with 0 (= one queue, rounds 2-4) / 8 / 24 / 32, then the chain trace
out=gpurun_out/${1:-r05m}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dataflow_protocol.py tests/test_gpu_headline_parity.py -x -q -m gpu 2>&1 | tail -15 > $out/gpu_tests.log; tail -3 $out/gpu_tests.log
B="python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
L=$PWD/gtsam_amd/lib
for rep in 1 2 3; do
  timeout 200 $B > $out/ab_default_$rep.json 2> $out/ab_default_$rep.err
  for v in x0 x8 x24 x32; do GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so timeout 200 $B > $out/ab_${v}_$rep.json 2> $out/ab_${v}_$rep.err; done
done
for v in default x0; do
  lib=$L/libgtsam_amd.so; [ $v = default ] || lib=$L/libgtsam_amd_$v.so
  GTSAM_AMD_LIB=$lib timeout 300 $B --workload venice1778 > $out/ab_${v}_venice1778.json 2> $out/ab_${v}_venice1778.err
done
GTG_DF_TRACE=1 timeout 300 python tools/df_trace.py --raw > $out/df_trace_default.txt 2> $out/df_trace_default.err
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/ab_*.json')):
    try:
        j = json.load(open(f)); ph = j['phase_ms_per_call']
        print(f.split('/')[-1], round(j['value'], 2), 'it/s', 'cholesky %.3f' % ph['cholesky'], '; error', repr(j['converged_error']))
    except Exception as e:
        print(f, 'failed', e, open(f.replace('.json', '.err')).read()[-400:])
PY
tail -2 $out/df_trace_default.txt | cut -c1-600
