"""pytest configuration: `gpu` marker, golden-fixture loader, optional live reference."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order (the driver runs `pytest -x`): parity files first, stress / protocol files last, so that a flake in a stress test can
# never again keep the parity tests from running (round 3: 117 of 119 GPU tests were not reached).
# Behind everything else: the tests whose subject is the round-1 stream / event schedule with one stream pair per part (GTG_CHOL=streams
# with a nested-dissection ordering -- an A/B and a last-resort fallback since round 3, not a default path).  Its `streams` variant of
# tests/test_gpu_parity.py::test_nested_dissection_schedules_are_equivalent did not return once on the last GPU session of round 4
# (profiles/r04_streams_tree_hang.txt); these tests run their device work in bounded child processes and come last, so that whatever
# they do, every other test has run.
_LAST = ("test_gpu_dataflow_protocol.py", "test_gpu_stress", "test_schedule_races.py")
_FIRST = ("test_gpu_parity.py", "test_gpu_headline_parity.py")
_VERY_LAST_FILES = ("test_gpu_speculative.py",)


def _order_key(item):
    f = os.path.basename(str(item.fspath))
    if f in _VERY_LAST_FILES or (f == "test_gpu_parity.py" and "[streams-" in item.name):
        return 3
    return 0 if f in _FIRST else (2 if any(f.startswith(x) for x in _LAST) else 1)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_order_key)   # (stable: the order inside a class of files is unchanged)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def live_ref():
    """The real reference (oracle/_ref), or None when the prebuilt .so did not travel."""
    from oracle import ref
    return ref if ref.available() else None
