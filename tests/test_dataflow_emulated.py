"""The dataflow Cholesky's two persistent kernels (csrc/chol_dataflow.hip: bulk_loop -- ticketed tile tasks, LDS-DMA staged contraction,
streamed substitution -- and chain_loop with chol_device.h::potrf_body inside) executed FROM THEIR OWN SOURCE on host threads
(tools/kernel_emu/dataflow_emu.cpp): one process per workgroup, one thread per work-item, device memory a shared mapping, so chain and
bulk workgroups run side by side and hand tiles over through the epoch-stamped flags exactly as on the device.  A small dense system
with its right-hand-side row: the factor and the forward solve against numpy -- the whole protocol (tickets, part / tile / pd flags,
pieces, accumulator lanes, slices, panel release) under timings no GPU produces; a dependency that is not really there shows as a hang
(the run is bounded) or as a wrong number.

The task lists are built here in the layout of chol_dataflow.hip::upload_df_plan, following build_df_plan_host's rules for one chain
(pieces of a contraction list, early pieces queued behind the block column of their youngest operand, lanes for tiles with many early
pieces); the library's own plan builder is pinned elsewhere (tests/test_chol_plan.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SRC = os.path.join(ROOT, "tools", "kernel_emu", "dataflow_emu.cpp")
EXE = os.path.join(ROOT, "tests", "_build", "dataflow_emu")
T = 128


def _build(path, flags):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the kernel emulator with")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    deps = [SRC, os.path.join(ROOT, "tools", "kernel_emu", "emu_hip.h"), os.path.join(ROOT, "gtsam_amd", "csrc", "chol_device.h"),
            os.path.join(ROOT, "gtsam_amd", "csrc", "chol_dataflow.hip")]
    if not os.path.exists(path) or any(os.path.getmtime(path) < os.path.getmtime(d) for d in deps):
        subprocess.run([CLANG, "-std=c++20", "-O2", "-pthread", "-Wno-psabi"] + flags + ["-o", path, SRC], check=True)
    return path


@pytest.fixture(scope="module")
def exe():
    return _build(EXE, [])


def plan_dense(nt, k_piece, k_final, lane_min=8, lane_max=8):
    """One chain over a dense lower triangle + the rhs row (tile row nt).  Slots: column by column."""
    slot = {}; n_slots = 0
    for J in range(nt):
        for I in range(J, nt + 1):
            slot[(I, J)] = n_slots; n_slots += 1
    finals = [[] for _ in range(nt)]; early = [[] for _ in range(nt)]
    klist = []
    for J in range(nt):
        for I in list(range(J, nt)) + [nt]:
            ks = list(range(0, J - 1)) if I == J else list(range(0, J))        # PD(J): block column J-1 is the chain workgroup's
            n = len(ks)
            m = max(0, n - k_final)
            while m > 0 and ks[m - 1] > J - 3:                                # an early piece sits two groups before its column at the latest
                m -= 1
            R = (m + k_piece - 1) // k_piece + 1
            off = len(klist)
            # a contraction step: the slots of its operand tiles (I, k), (J, k) and their 64-bit sub-tile masks (dense system: every 16 x 16
            # sub-tile; the right-hand-side row has one row of them), chol_dataflow.hip::kStepWords
            klist += [(slot[(I, k)], slot[(J, k)], 0xFF if I == nt else -1, 0 if I == nt else -1, -1, -1) for k in ks]
            gprev = 0
            for r in range(R - 1):
                b, e = r * k_piece, min(m, (r + 1) * k_piece)
                g = max(ks[e - 1] + 1, gprev)
                early[g].append([I, J, off + b, e - b, r, R]); gprev = g
            finals[J].append([I, J, off + m, n - m, R - 1, R])
    order = []
    for q in range(nt):
        order += finals[q]
        if q >= 1:
            order += early[q - 1]
    order += early[nt - 1]
    lanes = {}; n_scratch = 0
    for t in order:
        E = t[5] - 1
        if t[4] == t[5] - 1 and E >= lane_min and min(lane_max, E // 4) >= 2:
            G = min(lane_max, E // 4)
            lanes[(t[0], t[1])] = (G, n_slots + n_scratch); n_scratch += G - 1
    tasks = [t + [slot[(t[0], t[1])], slot[(t[1], t[1])]] + list(lanes.get((t[0], t[1]), (1, -1))) + ([0xFF, 0] if t[0] == nt else [-1, -1]) for t in order]   # [10, 11]: the tile's own sub-tile mask (dense; one row of sub-tiles in the right-hand-side row)
    chain_slots = []
    for J in range(nt):
        chain_slots += [slot[(J, J)], slot[(J, J - 1)] if J > 0 else -1, -1, -1]   # + the sub-tile mask of (J, J-1): dense
    t0 = list(range(0, nt, 2)); t1 = list(range(1, nt, 2))
    return dict(slot=slot, n_slots=n_slots, n_scratch=n_scratch, tasks=np.array(tasks, np.int32), klist=np.array(klist, np.int64).astype(np.int32).reshape(-1, 6),
                chain_slots=np.array(chain_slots, np.int32), chain_off=np.array([0, len(t0), nt], np.int32), chain_tiles=np.array(t0 + t1, np.int32),
                max_pieces=max(t[5] for t in order), lanes=lanes)


def factor(exe, nt, n_bulk, k_piece, k_final, tmp_path, seed=3, lane_min=8):
    P = plan_dense(nt, k_piece, k_final, lane_min=lane_min)
    rng = np.random.default_rng(seed)
    N = nt * T
    M = rng.standard_normal((N, N + 40)); A = M @ M.T + N * np.eye(N); g = rng.standard_normal(N)
    n_all = P["n_slots"] + P["n_scratch"]
    S = np.zeros((n_all, T, T))
    for J in range(nt):
        for I in range(J, nt):
            S[P["slot"][(I, J)]] = A[I * T:(I + 1) * T, J * T:(J + 1) * T]
        S[P["slot"][(nt, J)]][0, :] = g[J * T:(J + 1) * T]
    S[P["n_slots"]:] = np.nan                                  # scratch slots of the accumulator lanes: whatever was there before
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(np.array([nt, n_all, len(P["tasks"]), len(P["klist"]), 2, n_bulk, 4096], np.int64).tobytes())
        for a in (S, P["tasks"], P["klist"], P["chain_slots"], P["chain_off"], P["chain_tiles"]):
            f.write(np.ascontiguousarray(a).tobytes())
    r = subprocess.run([exe, fin, fout], timeout=1500)
    assert r.returncode == 0
    raw = open(fout, "rb").read()
    So = np.frombuffer(raw, np.float64, n_all * T * T, 0).reshape(n_all, T, T)
    o = So.nbytes + nt * T * T * 8
    fail = np.frombuffer(raw, np.float64, 2, o); ctrl = np.frombuffer(raw, np.int32, 16, o + 16)
    factor.last = (So.copy(), np.frombuffer(raw, np.float64, nt * T * T, So.nbytes).copy())
    assert fail[0] == 0.0 and fail[1] == 0.0, (fail, ctrl)
    assert ctrl[0] >= len(P["tasks"]) and ctrl[1] == nt        # every ticket taken, every diagonal tile factored
    L = np.linalg.cholesky(A); y = np.linalg.solve(L, g)
    for J in range(nt):
        for I in range(J, nt):
            blk = So[P["slot"][(I, J)]]
            assert np.abs((np.tril(blk) if I == J else blk) - L[I * T:(I + 1) * T, J * T:(J + 1) * T]).max() <= 1e-11 * np.abs(L).max(), (I, J)
        assert np.abs(So[P["slot"][(nt, J)]][0] - y[J * T:(J + 1) * T]).max() <= 1e-11 * max(1.0, np.abs(y).max()), J
    return P


def test_emulated_dataflow_factorisation_single_pieces(exe, tmp_path):
    factor(exe, 3, 2, 4, 4, tmp_path)


def test_emulated_dataflow_factorisation_with_early_pieces(exe, tmp_path):
    """Contraction lists cut into pieces of 2 steps with a final piece of 1: the partial results travel through the tiles and part_flag."""
    P = factor(exe, 6, 3, 2, 1, tmp_path)
    assert P["max_pieces"] >= 3 and not P["lanes"]


@pytest.mark.skipif(os.environ.get("GTG_TEST_SLOW") != "1", reason="three minutes of host threads: GTG_TEST_SLOW=1 runs it")
def test_emulated_dataflow_factorisation_with_accumulator_lanes(exe, tmp_path):
    """Pieces of one step on an 11-tile dense system: the tiles of the last block columns have >= 8 early pieces and accumulate in two
    lanes (scratch slots behind the stored tiles, NaN before the run), added up by the final piece in lane order."""
    P = factor(exe, 11, 4, 1, 1, tmp_path)
    assert P["lanes"] and P["n_scratch"] > 0


def test_emulated_dataflow_factorisation_is_reproducible_with_the_deferred_last_slice(exe, tmp_path):
    """The chain kernel applies the last slice of the tile left of a diagonal tile only to the blocks panel 0 reads; the rest runs under
    panel 0's pivot chain on the two wavefronts that idle there, before the barrier in front of the next panel's updates: every entry
    sees the same sums in the same order whatever the host threads' timing (the default since round 5; the numpy check is in `factor`)."""
    factor(exe, 3, 2, 2, 1, tmp_path)
    S0, X0 = factor.last
    for rep in range(2):
        factor(exe, 3, 2, 2, 1, tmp_path)
        S1, X1 = factor.last
        assert np.array_equal(S0, S1)
        assert np.array_equal(X0.reshape(3, T * T)[:, :14336], X1.reshape(3, T * T)[:, :14336])
