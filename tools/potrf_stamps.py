import numpy as np, ctypes as C, sys
sys.path.insert(0,'.')
from gtsam_amd import lib
from gtsam_amd.problem import Problem
L=lib.load()
dev=lib.DeviceGraph(Problem(var_type=np.array([0],np.int32)))
rng=np.random.default_rng(0); A=rng.normal(size=(128,200)); A=A@A.T+np.eye(128)
out=np.zeros(15,np.int64)
for it in range(3):
    B=A.copy()
    rc=L.gtg_debug_potrf_stamps(dev.h, B.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    d=np.diff(out)
    print(rc, "total cycles", out[14]-out[0], "load", d[0], "stages(potrf, inv/trsm, update)x4:", d[1:13].reshape(4,3).tolist(), "store", d[13])
print(np.abs(np.tril(B)-np.linalg.cholesky(A)).max())
