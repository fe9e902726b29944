"""The diagonal-tile body of both Cholesky schedules (csrc/chol_device.h::potrf_body: one pivot-chain wavefront, follower wavefronts
that replay its published columns a few pivots behind, deferred MFMA updates and write-back) executed from its OWN source on host
threads (tools/kernel_emu): the factor and the four 32 x 32 inverses against numpy, repetitions against each other bit for bit --
under thread timings no GPU produces, so a dependency that is only ever satisfied by the hardware's usual timing would show up as a
different bit or a hang -- and the failure flag of a tile that is not positive definite.

(Where lanes of ONE wavefront hand data to each other through LDS without a collective, the body relies on the hardware's in-order
LDS operations; those places carry GT_WAVE_SYNC(), which is nothing on the device and a rendezvous of the wavefront's threads here.)"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SRC = os.path.join(ROOT, "tools", "kernel_emu", "potrf_emu.cpp")
LIB = os.path.join(ROOT, "tests", "_build", "libpotrf_emu.so")


def _build(path, flags):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ to build the kernel emulator with")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    deps = [SRC, os.path.join(ROOT, "tools", "kernel_emu", "emu_hip.h"), os.path.join(ROOT, "gtsam_amd", "csrc", "chol_device.h")]
    if not os.path.exists(path) or any(os.path.getmtime(path) < os.path.getmtime(d) for d in deps):
        subprocess.run([CLANG, "-std=c++20", "-O2", "-pthread", "-fPIC", "-shared", "-Wno-psabi"] + flags + ["-o", path, SRC], check=True)
    lib = ctypes.CDLL(path)
    lib.emu_potrf128.restype = ctypes.c_longlong
    lib.emu_potrf128.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong]
    lib.emu_potrf128_ranktest.restype = ctypes.c_longlong
    lib.emu_potrf128_ranktest.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
    return lib


@pytest.fixture(scope="module")
def emu():
    return _build(LIB, [])


def _factor(lib, A, wt, epoch=1):
    tile = A.copy(); X = np.zeros(128 * 128); fail = np.zeros(2)
    flag = lib.emu_potrf128(tile.ctypes.data, X.ctypes.data, fail.ctypes.data, wt, epoch)
    return tile, X, fail, flag


@pytest.mark.parametrize("seed", [1, 2])
def test_emulated_diagonal_tile_body(emu, seed):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((128, 160)); A = M @ M.T + 64.0 * np.eye(128)
    L = np.linalg.cholesky(A)
    first = None
    for rep in range(4):
        wt = rep & 1                           # plain stores + release, then the write-through publication: the same numbers
        tile, X, fail, flag = _factor(emu, A, wt, epoch=3)
        assert flag == 3 * 8 + 4 and fail[0] == 0.0 and fail[1] == 0.0
        assert np.abs(np.tril(tile) - L).max() <= 1e-12 * np.abs(L).max()
        assert np.all(np.triu(tile[:32, :32], 1) == 0.0)          # the diagonal sub-blocks come back with their upper part zeroed
        for jb in range(4):
            Xi = X[1024 * jb:1024 * (jb + 1)].reshape(32, 32)
            assert np.abs(Xi - np.linalg.inv(L[32 * jb:32 * jb + 32, 32 * jb:32 * jb + 32])).max() <= 1e-12
        if first is None:
            first = (tile.copy(), X.copy())
        else:
            assert np.array_equal(np.tril(tile), np.tril(first[0])), rep
            assert np.array_equal(X[:4096], first[1][:4096]) and np.array_equal(X[4096:14336], first[1][4096:14336]), rep   # inverses, operand images


def test_emulated_body_flags_a_tile_that_is_not_positive_definite(emu):
    rng = np.random.default_rng(7)
    M = rng.standard_normal((128, 160)); A = M @ M.T + 64.0 * np.eye(128)
    A[70, 70] = -1.0
    _, _, fail, flag = _factor(emu, A, 1)
    assert fail[0] == 1.0 and flag == 8 + 4       # an error, never a hang: every panel is still released


def test_emulated_rank_test_of_the_reference(emu):
    """choleskyPartial's exponent test (base/cholesky.cpp:144-157) at the ends of the variables' pivot blocks: a tile of 9-dimensional
    variables passes when it is well conditioned and raises the flag when the last pivot of one variable is 2^-13 of the one before."""
    lib = emu
    kinds = np.zeros(128, np.uint8)
    kinds[8:126:9] = 1                                   # last pivot of every 9-dimensional variable
    rng = np.random.default_rng(9)
    M = rng.standard_normal((128, 160)); A = M @ M.T + 64.0 * np.eye(128)
    for bad in (False, True):
        B = A.copy()
        if bad:                                          # variable 3 (columns 27..35): its last direction almost unobserved
            L = np.linalg.cholesky(B); L[35, 35] *= 2.0 ** -14; B = L @ L.T
        tile = B.copy(); X = np.zeros(128 * 128); fail = np.zeros(2); texp = np.zeros(4)
        flag = lib.emu_potrf128_ranktest(tile.ctypes.data, X.ctypes.data, fail.ctypes.data, 1, 1, kinds.ctypes.data, texp.ctypes.data)
        assert flag == 8 + 4
        assert fail[0] == (1.0 if bad else 0.0), (bad, fail)
