#!/usr/bin/env python3
"""tools/time_sfm_bal.py [ladybug1723|dubrovnik16|venice1778] [--cpu-iterations N] -- the reference's benchmark program of the path
(timing/timeSFMBAL.cpp) with GpuLevenbergMarquardtOptimizer, host C++ end to end: writes the seeded synthetic problem as a BAL
file (native writer, gtg_io_write_bal = writeBAL's format), then runs tests/_build/time_sfm_bal_gpu on it (GTSAM's own loader,
GTSAM's own graph construction, extraction + upload + optimize() through the shim).  Prints the program's JSON line."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gtsam_amd import datasets as D  # noqa: E402
from gtsam_amd import io as IO  # noqa: E402


def main():
    name = next((a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()), "ladybug1723")
    gen = {"ladybug1723": D.ladybug_1723, "dubrovnik16": D.dubrovnik_16, "venice1778": D.venice_1778}[name]
    exe = os.path.join(ROOT, "tests", "_build", "time_sfm_bal_gpu")
    if not os.path.exists(exe):
        raise SystemExit("tests/_build/time_sfm_bal_gpu not built (make -C gtsam_amd/host, needs the GTSAM headers)")
    cams, pts, oc, op, oz = gen()
    path = os.path.join(tempfile.gettempdir(), f"gtsam_amd_{name}.txt")
    IO.write_bal(path, cams, pts, oc, op, oz)
    extra = []
    if "--cpu-iterations" in sys.argv:
        extra = ["--cpu-iterations", sys.argv[sys.argv.index("--cpu-iterations") + 1]]
    out = subprocess.run([exe, path] + extra, capture_output=True, text=True, timeout=3000)
    sys.stdout.write(out.stdout)
    if os.environ.get("GTG_DEBUG_TIMING"):
        sys.stderr.write(out.stderr)
    if out.returncode:
        sys.stderr.write(out.stderr)
        raise SystemExit(out.returncode)
    os.unlink(path)


if __name__ == "__main__":
    main()
