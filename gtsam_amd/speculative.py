"""Speculative lambda search over replicated handles: how the LM loop of this path uses more than one GPU when the factorisation's
critical path is serial.

LevenbergMarquardtOptimizer::iterate (nonlinear/LevenbergMarquardtOptimizer.cpp:273-308) linearises once and then walks through
lambda values until a step is accepted (tryLambda, :121-270); on a rejection the next value is known in advance --
lambda * factor, with the factor doubled unless useFixedLambdaFactor (internal/LevenbergMarquardtState.h:70-76).  Every try is a
full damp -> eliminate -> solve -> retract -> error pass (6.8 ms of the 8.9 ms iteration on the L1723 shape, 1.25 tries per
iteration), and the tries of one iteration are independent of each other: they share the linearisation and differ in lambda only.

So with N replicas (one process per GPU, every process holds the WHOLE graph, values identical everywhere):
  * every replica linearises (same values, same deterministic kernels: the same bits);
  * replica r tries the r-th lambda of the sequence the reference would walk through on consecutive rejections;
  * the N results (status + 4 scalars each) are gathered, and every process replays tryLambda's decisions IN ORDER on them -- the
    same code as the sequential host, fed from the gathered numbers instead of from a call: the first try that ends the search ends it;
  * if that try was accepted, its trial values are broadcast device-to-device from the replica that computed them into everybody's
    current values (2.7 MB on the L1723 shape); if no try of the round ended the search, the next round starts at the next lambda.
The trajectory (lambdas, errors, accept / reject decisions, values) is the sequential one bit for bit; an iteration costs one try
instead of 1.25 on the headline problem.  What it costs: the speculated tries that are thrown away (GPU time of otherwise idle
replicas) and one small gather + one broadcast per iteration.

Not used for graphs with smart factors: their triangulation cache is state that a discarded try would have touched in the
sequential order (SmartProjectionFactor.h:127-183), so the replicas fall back to lock-step sequential tries there.

The landmark-sharded mode of SURVEY.md section 8(e) (gtsam_amd/distributed.py: one exchange of the reduced camera system per try)
is the other way to use N GPUs; on this path it shortens the 1.4 ms in front of the factorisation and pays an all-reduce of 0.13 GB
for it, while the 5.2 ms factorisation -- a serial chain of 122 diagonal tiles -- stays replicated.  bench.py --parallelism
selects; the default for N > 1 is the landmark shard (the partition north_star names), this mode is reported beside it.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np

from .lib import GTG_INDETERMINATE
from .optimizer import DeviceLevenbergMarquardt


class TorchComm:
    """The two exchanges over torch.distributed: nccl (= RCCL over xGMI) on device tensors, gloo on host copies (CPU tests, or
    several replicas on one GPU).  The backend of the group decides; there is NO fall-back from one to the other: a device exchange
    that raises is an error on the rank that sees it (RCCL errors are asynchronous and per rank, so ranks could not agree on a
    switch of process groups without another collective that may hang in turn)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist; self.group = group
        self.rank = dist.get_rank(group); self.world = dist.get_world_size(group)
        self.device_backend = dist.get_backend(group) != "gloo"

    def all_gather(self, vec):
        import torch
        if self.device_backend:
            t = torch.tensor(np.asarray(vec, np.float64), dtype=torch.float64, device="cuda")
            out = torch.empty((self.world, t.numel()), dtype=torch.float64, device="cuda")
            self.dist.all_gather_into_tensor(out, t, group=self.group)
            return out.cpu().numpy()
        t = torch.tensor(np.asarray(vec, np.float64), dtype=torch.float64)
        out = torch.empty((self.world, t.numel()), dtype=torch.float64)
        self.dist.all_gather(list(out.unbind(0)), t, group=self.group)
        return out.numpy()

    def broadcast_accepted(self, dev, src):
        """The accepted trial values of replica `src` become everybody's current values."""
        import torch
        if self.rank == src:
            dev.accept()                           # trial <-> current: the winner's current values are now the accepted ones
        if self.device_backend:
            from .distributed import _DevicePtr
            ptr, n, stream = dev.values_device_ptr(0)
            t = torch.as_tensor(_DevicePtr(ptr, n), device="cuda")
            with torch.cuda.stream(torch.cuda.ExternalStream(int(stream))) if stream else _null():
                self.dist.broadcast(t, src=src, group=self.group)
            if self.rank != src:
                dev.values_changed()
            return
        if self.rank == src:
            t = torch.from_numpy(dev.values())
        else:
            t = torch.empty(dev.val_size, dtype=torch.float64)
        self.dist.broadcast(t, src=src, group=self.group)
        if self.rank != src:
            dev.set_values(t.numpy())


_RC_RAISED = -99   # status of a speculated try whose device call raised (travels through the gather)


class _null:
    def __enter__(self): return self
    def __exit__(self, *a): return False


class SpeculativeLevenbergMarquardt(DeviceLevenbergMarquardt):
    """DeviceLevenbergMarquardt over N replicas; same accessors, same trace."""

    def __init__(self, problem, values0, params=None, device=0, comm=None, reduced_ordering=None):
        super().__init__(problem, values0, params, device=device, reduced_ordering=reduced_ordering)
        self.comm = comm
        self.speculated = 0        # tries computed by this replica
        self.discarded = 0         # ... whose result the replay did not reach
        self._sequential = comm is None or comm.world == 1 or int(problem.n_smart) > 0 or self.params.linearSolverType == "Iterative"

    def _lambda_sequence(self, n):
        """The next n (lambda, factor-after) pairs of consecutive rejections (increaseLambda, LMState.h:70-76); None where the
        sequential search would have given up before trying (lambda >= lambdaUpperBound, LM.cpp:256-261)."""
        p = self.params
        lam, fac = self._lambda, self._factor
        seq = []
        for _ in range(n):
            seq.append(lam if (not seq or lam < p.lambdaUpperBound) else None)
            lam = lam * fac
            if not p.useFixedLambdaFactor:
                fac = fac * 2.0
        return seq

    def _try_lambda(self):
        if self._sequential:
            return super()._try_lambda()
        p = self.params
        comm = self.comm
        seq = self._lambda_sequence(comm.world)
        mine = seq[comm.rank]
        # A speculated try may fail where the sequential search would never have gone (a time-out of a lambda the trajectory does
        # not reach, any HIP error): the failure travels as a status code in the gathered vector -- every rank still takes part in
        # the collective -- and is raised, on ALL ranks, only if the in-order replay actually reaches that try.
        mine_error = None
        if mine is not None:
            try:
                rc, out = self.dev.try_lambda(mine, p.diagonalDamping, p.minDiagonal, p.maxDiagonal)
            except Exception as e:   # noqa: BLE001
                rc, out, mine_error = _RC_RAISED, np.zeros(4), e
            self.speculated += 1
        else:
            rc, out = -1, np.zeros(4)
        got = comm.all_gather([float(rc), out[0], out[1], out[2], out[3]])
        # replay tryLambda's decisions in order, the device call replaced by the gathered result of the replica that made it
        real_try = self.dev.try_lambda
        real_accept = self.dev.accept
        ended = False
        for k in range(comm.world):
            if seq[k] is None:
                break
            assert seq[k] == self._lambda, (seq, self._lambda)
            if int(got[k, 0]) == _RC_RAISED:          # the trajectory needs a try that failed on its replica: an error everywhere
                if k == comm.rank and mine_error is not None:
                    raise mine_error
                raise RuntimeError(f"speculative lambda search: the try of lambda = {seq[k]!r} failed on replica {k} (see that rank's error)")
            self.dev.try_lambda = lambda *a, _k=k: (int(got[_k, 0]), got[_k, 1:5].copy())
            accepted = {"yes": False}
            self.dev.accept = lambda: accepted.__setitem__("yes", True)
            try:
                ended = DeviceLevenbergMarquardt._try_lambda(self)
            finally:
                self.dev.try_lambda = real_try; self.dev.accept = real_accept
            if accepted["yes"]:
                comm.broadcast_accepted(self.dev, k)
            if ended:
                if mine is not None and comm.rank > k:
                    self.discarded += 1
                break
            if k + 1 < comm.world and seq[k + 1] is not None:
                self._write_log_file(self._error)        # (the reference logs after every try that keeps searching; iterate() logs the last one)
        return ended
