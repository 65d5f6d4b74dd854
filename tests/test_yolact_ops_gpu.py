"""Yolact Detect / postprocess HIP kernels vs the CPU oracle: indices, boxes, scores, masks all bit-exact."""
import numpy as np
import pytest

from oracle import ora

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,n,k", [(3, 1000, 100), (2, 19248, 200), (1, 201600, 1000), (5, 50, 100), (2, 4096, 1000), (1, 70000, 128), (2, 63000, 6000), (1, 9000, 8192),
                                      (2, 5000, 4819)])
def test_topk_matches_oracle(ffi, rows, n, k):
    rng = np.random.default_rng(rows * 1000 + n + k)
    keys = rng.uniform(0, 1, (rows, n)).astype(np.float32)
    # exact ties (1 %) and a block of identical values straddling the threshold
    ties = rng.integers(0, n, (rows, max(n // 100, 1)))
    for r in range(rows):
        keys[r, ties[r]] = keys[r, ties[r][0]]
    keys[0, : min(n, 3 * k) : 3] = 0.75
    vals, idx, cnt = ffi.topk(keys, k)
    for r in range(rows):
        s, i = ora.topk(keys[r], k)
        assert cnt[r] == len(s)
        assert np.array_equal(idx[r, : cnt[r]], i)
        assert np.array_equal(vals[r, : cnt[r]], s)


@pytest.mark.parametrize("rows,n,k,mode", [(8, 201600, 1000, "uniform"), (2, 50400, 1000, "ties"), (3, 40000, 600, "uniform"), (1, 100000, 1000, "equal"),
                                           (2, 131072, 1024, "few"), (4, 45000, 257, "neg"), (1, 400000, 1000, "ties")])
def test_topk_two_level_long_rows(ffi, rows, n, k, mode):
    """Few very long rows (the RPN pre-NMS top-k) go through slices: per-slice top-k, then top-k of the candidates.  Same total order
    (key descending, index ascending), ties across slice borders included."""
    rng = np.random.default_rng(n + k)
    if mode == "uniform":
        keys = rng.uniform(0, 1, (rows, n))
    elif mode == "ties":
        keys = rng.integers(0, 7, (rows, n)) / 8.0          # thousands of equal keys at the threshold, spread over every slice
    elif mode == "equal":
        keys = np.full((rows, n), 0.5)
    elif mode == "few":
        keys = np.zeros((rows, n)); keys[:, rng.integers(0, n, 300)] = rng.uniform(1, 2, 300)   # fewer distinct winners than k
    else:
        keys = -np.abs(rng.standard_normal((rows, n)))
    keys = keys.astype(np.float32)
    vals, idx, cnt = ffi.topk(keys, k)
    for r in range(rows):
        s, i = ora.topk(keys[r], k)
        assert cnt[r] == len(s) and np.array_equal(idx[r, : cnt[r]], i) and np.array_equal(vals[r, : cnt[r]], s)


def test_topk_all_equal_and_negative(ffi):
    keys = np.full((2, 5000), 0.5, np.float32)
    keys[1] = -np.arange(5000, dtype=np.float32)
    vals, idx, cnt = ffi.topk(keys, 200)
    assert np.array_equal(idx[0], np.arange(200)) and np.array_equal(idx[1], np.arange(200))
    assert np.array_equal(vals[1], -np.arange(200, dtype=np.float32))


@pytest.mark.parametrize("stride", [1, 2])
def test_deform_im2col_matches_oracle(ffi, stride):
    """DCNv2 sampling stage of the YOLACT++ backbones: bit-exact columns for random offsets (in range, out of range, exactly on
    the -1 / H borders, integer positions) and mask logits."""
    rng = np.random.default_rng(40 + stride)
    N, H, W, C = 2, 23, 31, 64
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    om = rng.normal(0, 3.0, (N, Ho, Wo, 27)).astype(np.float32)
    om[0, :4, :, :18] = np.round(om[0, :4, :, :18])      # integer sample positions
    om[1, 0, :, 0] = -float(0 * stride - 1 + 0) - 1.0     # tap 0 lands exactly on h = -1 (excluded)
    om[1, -1, :, 12] = float(H) - float((Ho - 1) * stride - 1 + 2)  # tap 6 lands exactly on h = H (excluded)
    got = ffi.deform_im2col(x, om, 3, 3, stride, 1, 1)
    ref = ora.deform_im2col(x, om, 3, 3, stride, 1, 1)
    assert got.shape == ref.shape == (N, Ho, Wo, 9 * C)
    assert np.array_equal(got, ref)
    assert (ref.reshape(-1, C) == 0).all(1).sum() > 50


def _yolact_inputs(rng, N, P, ncls=81, md=32, hot=0.02):
    conf = rng.standard_normal((N, P, ncls)).astype(np.float32)
    conf[..., 0] += 4.0  # background dominates, as in a real detector
    hotm = rng.uniform(0, 1, (N, P)) < hot
    cls = rng.integers(1, ncls, (N, P))
    boost = rng.uniform(3, 9, (N, P)).astype(np.float32)
    for n in range(N):
        idx = np.nonzero(hotm[n])[0]
        conf[n, idx, cls[n, idx]] += boost[n, idx]
    # clustered priors so that fast-NMS suppresses something
    ctr = rng.uniform(0.1, 0.9, (P, 2)); ctr[::2] = ctr[1::2][: len(ctr[::2])] if P % 2 == 0 else ctr[::2]
    priors = np.concatenate([ctr, rng.uniform(0.05, 0.4, (P, 2))], 1).astype(np.float32)
    loc = (rng.standard_normal((N, P, 4)) * 0.5).astype(np.float32)
    mask = np.tanh(rng.standard_normal((N, P, md))).astype(np.float32)
    return conf, loc, mask, priors


@pytest.mark.parametrize("N,P", [(2, 1500), (1, 19248), (3, 300)])
def test_yolact_detect_matches_oracle(ffi, N, P):
    rng = np.random.default_rng(N * 7 + P)
    conf, loc, mask, priors = _yolact_inputs(rng, N, P)
    got, boxes = ffi.yolact_detect(conf, loc, mask, priors)
    total = 0
    for n in range(N):
        ref_boxes = ora.yolact_decode(loc[n], priors)
        assert np.array_equal(boxes[n], ref_boxes)
        ref = ora.yolact_detect(ora.softmax(conf[n]), ref_boxes, mask[n])
        g = got[n]
        assert len(g["score"]) == len(ref["score"])
        for key in ("prior", "cls", "score", "box", "mask"):
            assert np.array_equal(g[key], ref[key]), key
        total += len(ref["score"])
    assert total > 0


def test_yolact_detect_empty(ffi):
    rng = np.random.default_rng(0)
    conf, loc, mask, priors = _yolact_inputs(rng, 2, 500, hot=0.0)
    conf[..., 0] += 20.0
    got, _ = ffi.yolact_detect(conf, loc, mask, priors)
    assert all(len(g["score"]) == 0 for g in got)


@pytest.mark.parametrize("h,w", [(138, 138), (550, 550), (97, 203)])
def test_yolact_masks_match_oracle(ffi, h, w):
    rng = np.random.default_rng(h + w)
    N, K, PH, PW = 2, 100, 138, 138
    proto = np.maximum(rng.standard_normal((N, PH, PW, 32)), 0).astype(np.float32)
    coeffs = np.tanh(rng.standard_normal((N, K, 32))).astype(np.float32)
    c = rng.uniform(0.1, 0.9, (N, K, 2)); s = rng.uniform(0.02, 0.5, (N, K, 2))
    boxes = np.concatenate([c - s / 2, c + s / 2], -1).astype(np.float32)
    boxes[0, 0] = [0.7, 0.8, 0.2, 0.1]  # swapped corners exercise sanitize()
    boxes[0, 1] = [-0.3, -0.2, 1.4, 1.2]  # beyond the image
    counts = np.array([37, 100], np.int32)
    masks, ib = ffi.yolact_masks(proto, coeffs, boxes, counts, h, w)
    for n in range(N):
        k = counts[n]
        ref_m, ref_b = ora.yolact_masks(proto[n], coeffs[n, :k], boxes[n, :k], h, w)
        assert np.array_equal(masks[n, :k], ref_m)
        assert np.array_equal(ib[n, :k], ref_b)
        assert ref_m.sum() > 0
