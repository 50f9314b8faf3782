import json
import logging
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    logging.getLogger("mct_quantizers_amd").setLevel(logging.ERROR)


@pytest.fixture(scope="session")
def golden_cases():
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(GOLDEN, "cases.npz"))
    return meta["cases"], arrays


@pytest.fixture(scope="session")
def half_cases():
    with open(os.path.join(GOLDEN, "cases_half.json")) as f:
        meta = json.load(f)
    arrays = np.load(os.path.join(GOLDEN, "cases_half.npz"))
    return meta["cases"], arrays


def finite_equal(got, want, x):
    """Bit equality wherever the input is finite (non-finite inputs are outside the parity domain:
    the reference's CPU path runs into int64-cast UB there)."""
    m = np.isfinite(np.asarray(x))
    got = np.ascontiguousarray(got, dtype=np.float32)
    want = np.ascontiguousarray(want, dtype=np.float32)
    return got.shape == want.shape and np.array_equal(got.view(np.uint32)[m], want.view(np.uint32)[m])


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    """Bit-for-bit float32 equality (distinguishes -0.0 from 0.0, compares NaN payloads)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def first_mismatch(a, b, x=None):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
    if a.shape != b.shape:
        return f"shape {a.shape} vs {b.shape}"
    bad = np.flatnonzero(a.view(np.uint32) != b.view(np.uint32))
    if bad.size == 0:
        return ""
    i = bad[0]
    xs = "" if x is None else f" x={np.asarray(x).reshape(-1)[i]!r}"
    return f"{bad.size} mismatches, first at {i}:{xs} got={a[i]!r} want={b[i]!r}"
