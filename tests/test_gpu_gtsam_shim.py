"""GPU: the C++ drop-in (gtsam_amd/host/GpuLevenbergMarquardtOptimizer, a subclass of
gtsam::LevenbergMarquardtOptimizer) against the reference's own optimizer on identical
NonlinearFactorGraph / Values / params -- tests/cpp/test_gpu_lm_gtsam.cpp, prebuilt in the build container
(needs GTSAM headers) together with oracle/_ref (the only libgtsam in this image)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_shim_matches_reference_optimizer():
    exe = os.path.join(ROOT, "tests", "_build", "test_gpu_lm_gtsam")
    if not os.path.exists(exe) or not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libgtsam_ref.so")):
        pytest.skip("prebuilt shim test / oracle/_ref did not travel")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print("\n".join(ln for ln in r.stdout.splitlines() if ln.startswith("FAIL")))     # (the failed expectations first: the output is long)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL PASSED" in r.stdout
