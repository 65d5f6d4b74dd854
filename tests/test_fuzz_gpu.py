"""Property/fuzz tests (hypothesis) of the selection kernels against the oracle: ragged sizes, ties, degenerate
boxes, all-equal scores -- the edge cases a fixed-seed test can miss.  Everything must match the oracle exactly."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import ora

pytestmark = pytest.mark.gpu
import os
_X = int(os.environ.get("ISEGMI_FUZZ_SCALE", "1"))   # ISEGMI_FUZZ_SCALE=20 python -m pytest tests/test_fuzz_gpu.py -m gpu: a long sweep (the round-6 one: gpurun_out/r6l_fuzz_long.txt)
SET = dict(max_examples=25 * _X, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


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


def _grid_boxes(rng, n, W=400, H=300):
    """integer boxes on a coarse grid: IoUs land on simple fractions -- exactly on 0.5 / 0.7 often enough to make `>` against `>=` and `+1` against plain areas matter"""
    x = rng.integers(0, W // 10, (n, 2)) * 10.0
    wh = rng.integers(1, 6, (n, 2)) * 10.0
    return np.concatenate([x, x + wh - 1], 1).astype(np.float32)


@settings(max_examples=20 * _X, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(R=st.integers(1, 1000), seed=st.integers(0, 2 ** 31), flags=st.integers(0, 7), kind=st.sampled_from(["random", "grid", "crowded"]), chip_wide=st.booleans(),
       dpi=st.sampled_from([5, 100]))
def test_box_postprocess_fuzz_under_every_fork(ffi, R, seed, flags, kind, chip_wide, dpi):
    """PostProcessor.filter_results under random fork flags (App. A.6: >=, plain areas, index order), ragged proposal counts, threshold-tie boxes and crowded classes
    (hundreds of candidates: the suppression matrix, in the block or on the whole chip) against the oracle with the same flags"""
    rng = np.random.default_rng(seed)
    ncls, N = 81, 2
    logits = rng.normal(0, 1.0, (N, R, ncls)).astype(np.float32)
    logits[..., 0] += 2.0
    hot = rng.choice(np.arange(1, ncls), 3, replace=False)
    logits[..., hot] += 4.0 if kind == "crowded" else 2.5
    if kind == "grid":
        props = np.stack([_grid_boxes(rng, R) for _ in range(N)])
        regr = np.zeros((N, R, 4 * ncls), np.float32)            # decode returns the proposal: the grid's exact IoUs survive
    else:
        c = rng.uniform(0, 1, (N, R, 2)) * [400, 300]; wh = np.exp(rng.uniform(np.log(8), np.log(200), (N, R, 2)))
        props = np.clip(np.concatenate([c - wh / 2, c + wh / 2], -1), 0, [399, 299, 399, 299]).astype(np.float32)
        regr = rng.normal(0, 0.3, (N, R, 4 * ncls)).astype(np.float32)
    cnt = np.array([R, rng.integers(0, R + 1)], np.int32)
    hw = np.array([[300, 400], [280, 390]], np.int32)
    got = ffi.box_postprocess(logits, regr, props, cnt, hw, det_per_img=dpi, nms_flags=flags, cap=dpi + 28, chip_wide=chip_wide)
    for n in range(N):
        k = int(cnt[n])
        rb, rs, rl = ora.box_postprocess(logits[n, :k], regr[n, :k], props[n, :k], float(hw[n, 1]), float(hw[n, 0]), det_per_img=dpi, nms_flags=flags, cap=dpi + 28)
        assert np.array_equal(got[n][2], rl) and np.array_equal(got[n][1], rs) and np.array_equal(got[n][0], rb), (n, k, flags)


@settings(max_examples=15 * _X, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(n=st.integers(1, 1000), seed=st.integers(0, 2 ** 31), flags=st.integers(0, 3), post=st.sampled_from([1, 50, 1000]), chip_wide=st.booleans(),
       min_size=st.sampled_from([0.0, 25.0]))
def test_rpn_level_fuzz_under_the_nms_forks(ffi, n, seed, flags, post, chip_wide, min_size):
    """one RPN level whose candidates ARE grid boxes (A = 1, zero deltas: decode returns the anchor), random scores with ties, under the NMS forks"""
    rng = np.random.default_rng(seed)
    boxes = _grid_boxes(rng, n)
    p = np.clip(rng.integers(1, 64, n) / 64.0, 0.02, 0.98).astype(np.float32)      # many equal scores: the index tie-break decides the visiting order
    head = np.zeros((1, n, 1, 5), np.float32)
    head[0, :, 0, 0] = np.log(p / (1 - p))
    hw = np.array([[300, 400]], np.int32)
    (gb, gs), = ffi.rpn_level(head, boxes, hw, 1, n, post, min_size=min_size, nms_flags=flags, chip_wide=chip_wide)
    rb, rs = ora.rpn_level(head[0, :, 0, 0], np.zeros((n, 4), np.float32), boxes, n, post, 0.7, min_size, 400.0, 300.0, flags)
    assert np.array_equal(gs, rs) and np.array_equal(gb, rb), (n, flags, post)


@settings(max_examples=12 * _X, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(K=st.integers(1, 300), seed=st.integers(0, 2 ** 31), aligned=st.booleans(), f16=st.booleans(), PH=st.sampled_from([7, 14]), nonfinite=st.booleans())
def test_roi_align_fuzz_table_plain_oracle(ffi, K, seed, aligned, f16, PH, nonfinite):
    """the three RoIAlign forms -- oracle, plain launch, table-driven launch -- on random RoIs (some overhanging, some thinner than a pixel), both sides of the
    `aligned` fork, fp32 and fp16 storage, with and without inf / NaN planted in row 0 / column 0 of every map"""
    rng = np.random.default_rng(seed)
    N, Cc = 2, 64
    dt = np.float16 if f16 else np.float32
    shapes = [(40, 60), (20, 30), (10, 15), (5, 8)]
    feats = [rng.standard_normal((N, h, w, Cc)).astype(dt) for h, w in shapes]
    if nonfinite:
        for f in feats:
            f[:, 0, ::2, :] = np.inf; f[:, 1:, 0, : Cc // 2] = np.nan
    scales = [0.25, 0.125, 0.0625, 0.03125]
    c = rng.uniform(-20, 260, (N, K, 2)) * [1.0, 0.7]; wh = np.exp(rng.uniform(np.log(0.5), np.log(300), (N, K, 2)))
    rois = np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32)
    counts = np.array([K, rng.integers(0, K + 1)], np.int32)
    ref = np.zeros((N, K, PH, PH, Cc), dt)
    for n in range(N):
        k = int(counts[n])
        lv = ora.level_map(rois[n, :k])
        for L in range(2, 6):
            idx = np.nonzero(lv == L)[0]
            if len(idx):
                r5 = np.concatenate([np.full((len(idx), 1), n, np.float32), rois[n, idx]], 1)
                ref[n, idx] = ora.roi_align(feats[L - 2].astype(np.float32), r5, scales[L - 2], PH, PH, 2, int(aligned)).astype(dt)
    plain = ffi.roi_align_f16(feats, scales, rois, counts, PH, PH, aligned=int(aligned)) if f16 else ffi.roi_align(feats, scales, rois, counts, PH, PH, aligned=int(aligned))[0]
    order, tab = ffi.roi_prep(rois, counts, shapes, scales, Cc, PH, PH, f16=f16, aligned=int(aligned))
    tabbed = ffi.roi_align_ordered(feats, scales, rois, counts, PH, PH, order, tab, f16=f16).reshape(ref.shape)
    for n in range(N):   # (rows past count: zero-filled by the plain launch, left as they were by the table-driven one)
        k = int(counts[n])
        assert np.array_equal(plain.reshape(ref.shape)[n, :k].astype(np.float32), ref[n, :k].astype(np.float32), equal_nan=True)
        assert np.array_equal(tabbed[n, :k].astype(np.float32), ref[n, :k].astype(np.float32), equal_nan=True)
