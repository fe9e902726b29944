import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
from gtsam_amd import lib
from gtsam_amd.problem import Problem
L = lib.load(); L.gtg_debug_syrk_ms.restype = C.c_double
dev = lib.DeviceGraph(Problem(var_type=np.array([0], np.int32)))
m = 100
tiles = m * (m + 1) // 2
flops = tiles * 2.0 * 128 * 128 * 256
for abl, name in ((0, "full"), (1, "no DMA"), (3, "no DMA, no C load"), (6, "no C load/store"), (7, "no DMA/C load/C store"), (15, "no DMA/C/MFMA (LDS reads + VALU only)")):
    L.gtg_debug_syrk_ms(dev.h, m, abl, 1)
    ms = L.gtg_debug_syrk_ms(dev.h, m, abl, 5)
    print(f"{name:42s} {ms:8.3f} ms  {flops / ms / 1e9:7.2f} TFLOP/s-equivalent")
