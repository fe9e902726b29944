#!/bin/bash
# tools/r05_session6.sh [tag] -- restored Cholesky kernels + k_build_diag / offset changes: parity, bench x2 (+ records A/B), other workloads, kernel stats
out=gpurun_out/${1:-r05f}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_parity.py -x -q -m gpu 2>&1 | tail -5 > $out/tests.log; tail -1 $out/tests.log
B="python bench.py --steps 16 --warmup 4 --cpu-baseline off --skip-dense-roofline --traffic off --host python"
L=$PWD/gtsam_amd/lib
for rep in 1 2; do
  for v in default records; do
    if [ $v = default ]; then unset GTSAM_AMD_LIB; else export GTSAM_AMD_LIB=$L/libgtsam_amd_$v.so; fi
    timeout 200 $B > $out/ab_${v}_$rep.json 2> $out/ab_${v}_$rep.err
  done
done
unset GTSAM_AMD_LIB
for w in venice1778 sphere2500 w20000 dubrovnik16; do timeout 300 $B --workload $w > $out/ab_default_$w.json 2> $out/ab_default_$w.err; done
python - <<PY
import json, glob
for f in sorted(glob.glob('$out/ab_*.json')):
    try:
        j = json.load(open(f)); ph = j['phase_ms_per_call']
        print(f.split('/')[-1], round(j['value'], 2), 'it/s', round(j['lambda_tries_per_s'], 2), 'tries/s;', ' '.join('%s %.3f' % (k, v) for k, v in ph.items()), '; error', repr(j['converged_error']))
    except Exception as e:
        print(f, 'failed', e, open(f.replace('.json', '.err')).read()[-400:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_r05f -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --cpu-baseline off --skip-dense-roofline --traffic off --host python > $GRAFT_REPO_ROOT/$out/prof_bench.json 2> $GRAFT_REPO_ROOT/$out/prof_bench.err
cd $GRAFT_REPO_ROOT
python tools/rocprof_top.py $(find /tmp/prof_r05f -name "*.db" | head -1) $out/kernel_stats.csv > /dev/null 2> $out/kernel_stats.err
head -22 $out/kernel_stats.csv | cut -c1-60,200-
