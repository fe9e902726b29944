# one GPU session: $1 = output directory under gpurun_out
out=gpurun_out/$1; mkdir -p $out
run() { echo "== $*" ; timeout 900 "${@}" 2>&1 | grep -v amdgpu.ids; }
{
run python -m pytest tests/test_gpu_parity.py -x -q -k "nested or sphere2500 or w20000 or orderings or backward_sweep or two_handles"
for w in sphere2500 w20000; do run python bench.py --cpu-baseline off --skip-dense-roofline --workload $w; done
GTG_DF_SLOTS=8 run python bench.py --cpu-baseline off --skip-dense-roofline --workload sphere2500
GTG_DF_SLOTS=8 run python bench.py --cpu-baseline off --skip-dense-roofline --workload w20000
GTG_DEBUG_TIMING=1 run python -c "
import sys, time
sys.path.insert(0, '.')
from tools import host_profile as HP
from gtsam_amd import lib as L
p, _ = HP.problem_for('w20000')
for i in range(2):
    t = time.time(); g = L.DeviceGraph(p); print('construct', time.time() - t, flush=True)
"
} > $out/log.txt 2>&1
cut -c1-700 $out/log.txt
