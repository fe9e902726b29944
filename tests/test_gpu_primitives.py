"""The hand-written primitives of the set-up passes (gtsam_amd/csrc/primitives.hip: exclusive scan, stable LSD radix sort, runs of a
sorted array -- rocPRIM's place until round 5) against numpy, through the C ABI's debug entry points; sizes around the tile boundaries."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 63, 64, 255, 256, 2047, 2048, 2049, 4095, 4096, 4097, 65536 + 17, 1_000_003]


def _lib():
    from gtsam_amd.lib import load
    return load()


@pytest.mark.parametrize("n", SIZES)
def test_exclusive_scan(n):
    rng = np.random.default_rng(n)
    a = rng.integers(-5, 1 << 33, n).astype(np.int64)
    out = np.zeros(n, np.int64)
    assert _lib().gtg_debug_scan(0, a.ctypes.data, n, out.ctypes.data) == 0
    want = np.concatenate([[0], np.cumsum(a)[:-1]])
    assert np.array_equal(out, want)


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("key_bytes,bits", [(8, 1), (8, 22), (8, 33), (8, 42), (8, 64), (4, 5), (4, 18), (4, 32)])
def test_stable_radix_sort_pairs(n, key_bytes, bits):
    rng = np.random.default_rng(n * 131 + bits)
    dt = np.uint64 if key_bytes == 8 else np.uint32
    # few distinct keys in the sorted bits (stability matters), garbage above them (must be ignored by the ordering, kept in the keys)
    low = rng.integers(0, min(1 << min(bits, 62), max(2, n // 7)), n, dtype=np.uint64)
    high = rng.integers(0, 1 << 16, n, dtype=np.uint64) << np.uint64(bits) if bits < 8 * key_bytes - 16 else np.zeros(n, np.uint64)
    key = (low | high).astype(dt)
    val = np.arange(n, dtype=np.uint32)
    ko = np.zeros(n, dt); vo = np.zeros(n, np.uint32)
    assert _lib().gtg_debug_sort_pairs(0, key_bytes, key.ctypes.data, val.ctypes.data, n, bits, ko.ctypes.data, vo.ctypes.data) == 0
    mask = np.uint64((1 << bits) - 1) if bits < 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    order = np.argsort(key.astype(np.uint64) & mask, kind="stable")
    assert np.array_equal(vo, val[order]) and np.array_equal(ko, key[order])


@pytest.mark.parametrize("n", [1, 4096, 100_001])
def test_radix_sort_keys_only_and_runs(n):
    rng = np.random.default_rng(n)
    key = rng.integers(0, max(2, n // 3), n, dtype=np.uint64) | (rng.integers(0, 1500, n, dtype=np.uint64) << np.uint64(32))
    ko = np.zeros(n, np.uint64)
    assert _lib().gtg_debug_sort_pairs(0, 8, key.ctypes.data, None, n, 44, ko.ctypes.data, None) == 0
    assert np.array_equal(ko, np.sort(key))
    uniq = np.zeros(n, np.uint64); start = np.zeros(n + 1, np.int64); nr = np.zeros(1, np.int32)
    assert _lib().gtg_debug_runs(0, ko.ctypes.data, n, uniq.ctypes.data, start.ctypes.data, nr.ctypes.data) == 0
    u, first = np.unique(ko, return_index=True)
    assert nr[0] == u.size and np.array_equal(uniq[:nr[0]], u) and np.array_equal(start[:nr[0]], first) and start[nr[0]] == n
    uniq2 = np.zeros(n, np.uint64); nr2 = np.zeros(1, np.int32)
    assert _lib().gtg_debug_runs(0, ko.ctypes.data, n, uniq2.ctypes.data, None, nr2.ctypes.data) == 0
    assert nr2[0] == u.size and np.array_equal(uniq2[:nr2[0]], u)
