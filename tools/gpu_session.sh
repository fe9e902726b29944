# one GPU session: $1 = output directory under gpurun_out
out=gpurun_out/$1; mkdir -p $out
run() { echo "== $*" ; timeout 900 "${@}" 2>&1 | grep -v amdgpu.ids; }
{
run python -m pytest tests/test_gpu_gtsam_shim.py tests/test_gpu_parity.py -x -q
GTG_DEBUG_TIMING=1 run python tools/time_sfm_bal.py ladybug1723
run python bench.py --skip-dense-roofline --workload streets1723
GTG_ORDERING=auto run python bench.py --cpu-baseline off --skip-dense-roofline --workload streets1723
GTG_DEBUG_TIMING=1 run python -c "
import sys, time
sys.path.insert(0, '.')
import numpy as np
from gtsam_amd import lib as L
from tests import problems as PB
from tests.conftest import load_golden
p, v0 = PB.dubrovnik_timesfm(load_golden('dubrovnik_3_7'))
for i in range(4):
    t = time.perf_counter(); g = L.DeviceGraph(p); t1 = time.perf_counter(); g.set_values(v0); e = g.error(); t2 = time.perf_counter(); g.linearize(); g.try_lambda(1e-3, True); t3 = time.perf_counter(); g.close(); t4 = time.perf_counter()
    print('small graph: construct %.2f ms, set_values + error %.2f ms, linearize + try_lambda %.2f ms, close %.2f ms' % (1e3*(t1-t), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)), flush=True)
"
} > $out/log.txt 2>&1
cut -c1-1200 $out/log.txt
