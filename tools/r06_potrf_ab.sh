#!/bin/bash
# tools/r06_potrf_ab.sh <tag> <variant ...> -- diagonal-tile variants (make variant NAME=..): per-stage stamps and the Cholesky time of the quick bench, default first
out=gpurun_out/${1:-r06n}; mkdir -p $out; shift
for v in default "$@"; do
  if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$PWD/gtsam_amd/lib/libgtsam_amd_$v.so; fi
  python tools/df_potrf_stamps.py ladybug1723 2>/dev/null > $out/stamps_$v.json
  for rep in 1 2; do
  timeout 600 python bench.py --workloads off --cpu-baseline off --traffic off --skip-dense-roofline --host python > $out/b.json 2> $out/b.err
  python - <<PY
import json
j=json.loads([l for l in open('$out/b.json') if l.startswith('{')][-1]); s=json.load(open('$out/stamps_$v.json'))
print('$v', 'chol ms', round(j['roofline']['ms_per_launch'],4), 'it/s', round(j['value'],2), 'err', j['converged_error'], 'potrf_body', round(s['potrf_body_us_median'],2), 'stages', [round(x,2) for x in s['stages_us_median'].values()]); print('   waves', s.get('wavefront_done_after_stage_start_us_median')); print('   p2 followers', [v for k, v in s.items() if k.startswith('panel 2 followers')])
PY
  done
done | tee $out/potrf_ab.txt
