"""Property/fuzz tests (hypothesis) of the selection kernels against the oracle: ragged sizes, ties, degenerate
boxes, all-equal scores -- the edge cases a fixed-seed test can miss.  Everything must match the oracle exactly."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import ora

pytestmark = pytest.mark.gpu
SET = dict(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


@settings(**SET)
@given(n=st.integers(1, 3000), k=st.sampled_from([1, 7, 100, 128, 200, 1000]), seed=st.integers(0, 2 ** 31), mode=st.sampled_from(["uniform", "ties", "equal", "neg"]))
def test_topk_fuzz(ffi, n, k, seed, mode):
    rng = np.random.default_rng(seed)
    if mode == "uniform":
        keys = rng.uniform(0, 1, (2, n))
    elif mode == "ties":
        keys = rng.integers(0, 5, (2, n)) / 4.0
    elif mode == "equal":
        keys = np.full((2, n), 0.25)
    else:
        keys = rng.standard_normal((2, n))
    keys = keys.astype(np.float32)
    vals, idx, cnt = ffi.topk(keys, k)
    for r in range(2):
        s, i = ora.topk(keys[r], k)
        assert cnt[r] == len(s) and np.array_equal(idx[r, : cnt[r]], i) and np.array_equal(vals[r, : cnt[r]], s)


@settings(**SET)
@given(n=st.integers(1, 1024), seed=st.integers(0, 2 ** 31), thr=st.sampled_from([0.3, 0.5, 0.7]), plus_one=st.booleans(), ge=st.booleans(),
       kind=st.sampled_from(["random", "grid", "identical", "degenerate"]), max_keep=st.sampled_from([0, 1, 50]))
def test_nms_fuzz(ffi, n, seed, thr, plus_one, ge, kind, max_keep):
    rng = np.random.default_rng(seed)
    if kind == "random":
        c = rng.uniform(0, 300, (n, 2)); wh = rng.uniform(1, 120, (n, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    elif kind == "grid":       # integer boxes: many IoUs land exactly on simple fractions (threshold ties)
        x = rng.integers(0, 12, (n, 2)) * 5.0; wh = rng.integers(1, 5, (n, 2)) * 5.0
        b = np.concatenate([x, x + wh - 1], 1)
    elif kind == "identical":
        b = np.tile(np.array([[10, 10, 50, 60]], np.float64), (n, 1))
    else:                      # zero / negative extents
        x = rng.uniform(0, 100, (n, 2)); b = np.concatenate([x, x - rng.integers(0, 2, (n, 2))], 1)
    b = b.astype(np.float32)
    s = (rng.integers(0, 8, n) / 8.0 if seed % 2 else rng.uniform(0, 1, n)).astype(np.float32)
    got = ffi.nms(b[None], s[None], thr, int(plus_one), int(ge), max_keep)[0]
    ref = ora.nms(b, s, thr, int(plus_one), int(ge), max_keep)
    assert np.array_equal(got, ref)
