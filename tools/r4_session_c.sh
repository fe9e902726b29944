#!/bin/bash
# tools/r4_session_c.sh -- round 4, third GPU session: the reduced system stored by tiles (context.h::SMat).  GPU suite, the bench line,
# the other workloads (memory per handle, set-up), a short stress.
out=gpurun_out/r4c; mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $out/gpu_tests.log
tail -3 $out/gpu_tests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
for w in sphere2500 w20000 venice1778 dubrovnik16; do
  timeout 600 python bench.py --workload $w --cpu-baseline off --skip-dense-roofline --traffic off > $out/bench_$w.json 2> $out/bench_$w.err
done
timeout 300 python tools/df_stress.py 100 3 > $out/stress.txt 2> $out/stress.err
tail -1 $out/stress.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c/bench*.json')):
    try:
        j=json.load(open(f)); print(f, round(j['value'],2), round(j['ms_per_step'],3), j.get('device_memory_per_handle_bytes'), round(j['time_to_converged_setup_s'],4), {k:round(v,3) for k,v in j['phase_ms_per_call'].items()})
    except Exception as e: print(f, 'failed', e)
PY
