"""Wrapper used by tests/test_bench_dry_run.py: runs bench.py's control flow WITHOUT a GPU -- under tools/hipstub (kernels do
not run, every number is meaningless) with torch.cuda patched out and the `nccl` process group replaced by `gloo` -- to
exercise exactly the code the driver runs for N = 1, 2, 4, 8: argument handling, sharded set-up with the layout check
through the real all-reduce callback, the lock-step iteration loop, barriers, the max-over-ranks reduction, rank 0's JSON."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.mem_get_info = lambda *a, **k: (0, 0)
_init = dist.init_process_group


def _init_gloo(backend, **kw):
    kw.pop("device_id", None)
    return _init("gloo", **kw)


dist.init_process_group = _init_gloo
_tensor = torch.tensor


def _cpu_tensor(*a, **kw):
    kw.pop("device", None)
    return _tensor(*a, **kw)


torch.tensor = _cpu_tensor
sys.argv = ["bench.py", "--gpus", os.environ.get("WORLD_SIZE", "1"), "--steps", "2", "--warmup", "1", "--workload", "dubrovnik16",
            "--cpu-baseline", "off", "--skip-dense-roofline", "--traffic", "off", "--host", "python",
            "--parallelism", os.environ.get("BENCH_PARALLELISM", "shard")]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
