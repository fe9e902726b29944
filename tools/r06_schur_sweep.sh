#!/bin/bash
# tools/r06_schur_sweep.sh [tag] -- k_schur_pairs against its occupancy (GTG_SCHUR_PAD_KB: untouched dynamic LDS per workgroup) and the
# XCD-contiguous block order (GTG_SCHUR_XCD=1): Schur phase per lambda try on L1723 and Venice
REPO=${GRAFT_REPO_ROOT:-$PWD}
out=$REPO/gpurun_out/${1:-r12a}; mkdir -p $out
cd $REPO
B="--steps 8 --warmup 2 --cpu-baseline off --skip-dense-roofline --traffic off --host python --workloads off"
for x in 0 1; do for pad in 0 16 37 64 100; do
  E="GTG_SCHUR_PAD_KB=$pad"; [ $x = 1 ] && E="$E GTG_SCHUR_XCD=1"
  for w in ladybug1723 venice1778; do
    env $E timeout 300 python bench.py --workload $w $B > $out/${w}_x${x}_p${pad}.json 2> $out/${w}_x${x}_p${pad}.err
  done
  python - <<PY
import json
r=[]
for w in ('ladybug1723','venice1778'):
    try:
        j=json.loads([l for l in open('$out/%s_x${x}_p${pad}.json' % w) if l.startswith('{')][-1]); r.append('%s schur %.3f ms (%.1f it/s)' % (w, j['phase_ms_per_call']['schur'], j['value']))
    except Exception as e: r.append('%s failed %s' % (w, e))
print('xcd $x pad $pad KB:', '; '.join(r))
PY
done; done
