import sys, time
sys.path.insert(0, '.')
import numpy as np
from gtsam_amd import datasets as D
from gtsam_amd.problem import bal_problem
from gtsam_amd import lib
from gtsam_amd.optimizer import DeviceLevenbergMarquardt
from gtsam_amd.params import LevenbergMarquardtParams as LMP
t = time.perf_counter(); p, v0 = bal_problem(*D.ladybug_1723()); print("generate problem", time.perf_counter() - t)
lib.load()
for rep in range(2):
    t = time.perf_counter(); dev = lib.DeviceGraph(p); t1 = time.perf_counter()
    dev.set_values(v0); e = dev.error(); t2 = time.perf_counter()
    print(f"rep {rep}: create+upload+analyze {t1 - t:.3f} s, set_values+error {t2 - t1:.4f} s")
    dev.close()
t = time.perf_counter(); opt = DeviceLevenbergMarquardt(p, v0, LMP.CeresDefaults()); t1 = time.perf_counter(); opt.optimize(); t2 = time.perf_counter()
print(f"construct {t1 - t:.3f} s, optimize {t2 - t1:.3f} s, iterations {opt.iterations()} inner {opt.getInnerIterations()} error {opt.error():.6f}")
