#!/bin/bash
# tools/r06_schur_variants.sh [tag] -- Schur phase of A/B builds of assemble.hip (GT_SCHUR_WIDE loads in flight per wavefront; GT_SCHUR_SPLIT: two accumulators)
REPO=${GRAFT_REPO_ROOT:-$PWD}
out=$REPO/gpurun_out/${1:-r12c}; mkdir -p $out
cd $REPO
B="--steps 8 --warmup 2 --cpu-baseline off --skip-dense-roofline --traffic off --host python --workloads off"
for rep in 1 2; do for v in "" _w6 _w12 _w16 _wsplit; do
  for w in ladybug1723 venice1778; do
    GTSAM_AMD_LIB=$REPO/gtsam_amd/lib/libgtsam_amd$v.so timeout 300 python bench.py --workload $w $B > $out/${w}$v.json 2> $out/${w}$v.err
  done
  python - <<PY
import json
r=[]
for w in ('ladybug1723','venice1778'):
    try:
        j=json.loads([l for l in open('$out/%s$v.json' % w) if l.startswith('{')][-1]); r.append('%s schur %.3f ms (%.1f it/s)' % (w, j['phase_ms_per_call']['schur'], j['value']))
    except Exception as e: r.append('%s failed %s' % (w, e))
print('variant "$v":', '; '.join(r))
PY
done; done
