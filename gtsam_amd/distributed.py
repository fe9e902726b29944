"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The data path has exactly one exchange step per lambda try: every shard owns a
subset of the landmarks with all their observation factors, builds its partial reduced camera system
[S | g] and the library calls back here to all-reduce (sum) it; plus three tiny exchanges (Hessian
diagonal for the damping, the landmark part of delta, a handful of scalars).  torch is plumbing only:
it wraps the library's DEVICE pointer in a tensor (no copy) and runs the collective.
"""
from __future__ import annotations

import ctypes

import numpy as np


class _DevicePtr:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can view it zero-copy."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(n),), "typestr": "<f8",
                                         "version": 2, "strides": None}


def make_allreduce(group=None):
    """Returns fn(ptr, n_doubles, stream) summing the buffer in place across the process group.
    Works for CUDA/HIP pointers (nccl/RCCL) and, for the CPU gloo tests, for host pointers."""
    import torch
    import torch.distributed as dist

    def fn(ptr, n, stream):
        backend = dist.get_backend(group)
        if backend == "gloo":
            buf = (ctypes.c_double * n).from_address(ptr)
            t = torch.from_numpy(np.frombuffer(buf, dtype=np.float64))
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return
        # Stream-ordered, no host round trip: the library hands over the stream its kernels are on; the collective is
        # enqueued with that stream current, so RCCL's stream waits for what is queued on it and the stream waits for the
        # collective (c10d's stream semantics) -- the library's next kernels follow in order.  (Four exchanges per lambda
        # try: two device-wide synchronisations around each were 8 host round trips on the critical path.)
        t = torch.as_tensor(_DevicePtr(ptr, n), device="cuda")
        if stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream))):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        else:
            torch.cuda.synchronize()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            torch.cuda.synchronize()

    return fn


def shard_of_landmark(rank_among_points: int, n_shards: int) -> int:
    """Ownership rule shared with the library (include/gtsam_amd.h gtg_upload_problem)."""
    return rank_among_points % n_shards
