"""Mask R-CNN RoI kernels vs the CPU oracle: indices, boxes, scores, features, masks all bit-exact."""
import numpy as np
import pytest

from oracle import ora

pytestmark = pytest.mark.gpu


def _boxes(rng, n, W=1333, H=800, clustered=True):
    c = rng.uniform(0, 1, (n, 2)) * [W, H]
    if clustered:
        c[n // 2:] = c[: n - n // 2] + rng.normal(0, 6, (n - n // 2, 2))
    wh = np.exp(rng.uniform(np.log(16), np.log(512), (n, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    return np.clip(b, 0, [W - 1, H - 1, W - 1, H - 1]).astype(np.float32)


@pytest.mark.parametrize("n", [3, 64, 200, 1000, 1025, 4819, 6000])  # SURVEY 8d unit sizes; > 1024 takes the 6144-box LDS kernel
@pytest.mark.parametrize("plus_one,ge", [(1, 0), (1, 1), (0, 0)])
def test_nms_matches_oracle(ffi, n, plus_one, ge):
    rng = np.random.default_rng(n + plus_one * 10 + ge)
    P = 4 if n <= 1024 else 2
    boxes = np.stack([_boxes(rng, n) for _ in range(P)])
    scores = rng.uniform(0, 1, (P, n)).astype(np.float32)
    scores[:, ::7] = scores[:, :1]  # ties
    for thr, mk in ((0.7, 0), (0.5, 0), (0.5, max(1, n // 3))):
        got = ffi.nms(boxes, scores, thr, plus_one, ge, mk)
        for p in range(P):
            ref = ora.nms(boxes[p], scores[p], thr, plus_one, ge, mk)
            assert np.array_equal(got[p], ref), (p, thr, mk)


def test_nms_known_answers(ffi):
    # App. A.6 KATs: identical boxes; IoU exactly == thr; chain where a suppressed box must not suppress.
    b = np.array([[[0, 0, 9, 9], [0, 0, 9, 9], [0, 0, 9, 4], [100, 100, 120, 120]]], np.float32)
    s = np.array([[0.9, 0.8, 0.7, 0.6]], np.float32)
    assert list(ffi.nms(b, s, 0.5, 1, 0)[0]) == [0, 2, 3]  # iou(0,2) = 50/100 = 0.5, not > 0.5
    assert list(ffi.nms(b, s, 0.5, 1, 1)[0]) == [0, 3]     # >= suppresses it
    chain = np.array([[[0, 0, 99, 99], [0, 0, 99, 59], [0, 0, 99, 35]]], np.float32)  # A-B .6, B-C .6, A-C .36
    assert list(ffi.nms(chain, np.array([[0.9, 0.8, 0.7]], np.float32), 0.5, 1, 0)[0]) == [0, 2]


def test_roi_align_matches_oracle(ffi):
    rng = np.random.default_rng(11)
    N, K, Cc = 2, 150, 64
    shapes = [(50, 84), (25, 42), (13, 21), (7, 11)]
    feats = [rng.standard_normal((N, h, w, Cc)).astype(np.float32) for h, w in shapes]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    rois = np.stack([_boxes(rng, K, 336, 200, False) for _ in range(N)])
    rois[0, 0] = [-50, -40, -10, -5]        # fully outside -> zeros
    rois[0, 1] = [10, 10, 10.2, 10.1]       # tiny -> roi size clamp to 1
    rois[1, 2] = [0, 0, 335, 199]
    counts = np.array([K, 97], np.int32)
    for PH in (7, 14):
        out, lv = ffi.roi_align(feats, scales, rois, counts, PH, PH)
        out = out.reshape(N, K, PH, PH, Cc)
        for n in range(N):
            k = counts[n]
            ref_lv = ora.level_map(rois[n, :k])
            assert np.array_equal(lv[n, :k], ref_lv)
            for L in range(2, 6):
                idx = np.nonzero(ref_lv == L)[0]
                if len(idx) == 0:
                    continue
                r5 = np.concatenate([np.full((len(idx), 1), n, np.float32), rois[n, idx]], 1)
                ref = ora.roi_align(feats[L - 2], r5, scales[L - 2], PH, PH, 2)
                assert np.array_equal(out[n, idx], ref), (PH, n, L)
            assert not out[n, k:].any()
    # known answers: constant map -> constant; ramp f[y,x]=x -> sample-x mean
    const = [np.full((1, 20, 30, 4), 3.5, np.float32)]
    o, _ = ffi.roi_align(const, [0.5], np.array([[[4, 6, 30, 28]]], np.float32), np.array([1], np.int32), 7, 7, fixed_level=0)
    assert np.allclose(o, 3.5, rtol=0, atol=2e-6)  # bilinear weights sum to 1 only up to rounding


def test_roi_align_adaptive_sampling_and_avgpool_match_oracle(ffi):
    """sampling_ratio = 0 (adaptive ceil(roi/pooled) grid; the R-50-C4 pooler) on one stride-16 map, and the C4 head's
    whole-window average pool."""
    rng = np.random.default_rng(21)
    N, K, Cc = 2, 80, 32
    feat = rng.standard_normal((N, 50, 84, Cc)).astype(np.float32)
    rois = np.stack([_boxes(rng, K, 1344, 800, False) for _ in range(N)])
    rois[0, 0] = [0, 0, 1343, 799]          # 6 x 4 samples per bin at 14 x 14
    rois[0, 1] = [100, 100, 103, 102]       # tiny -> 1 x 1
    counts = np.array([K, 33], np.int32)
    for PH in (14, 7):
        out, _ = ffi.roi_align([feat], [1.0 / 16], rois, counts, PH, PH, sampling=0, fixed_level=0)
        out = out.reshape(N, K, PH, PH, Cc)
        for n in range(N):
            k = counts[n]
            r5 = np.concatenate([np.full((k, 1), n, np.float32), rois[n, :k]], 1)
            ref = ora.roi_align(feat, r5, 1.0 / 16, PH, PH, 0)
            assert np.array_equal(out[n, :k], ref), (PH, n)
            assert not out[n, k:].any()
    x = rng.standard_normal((37, 7, 7, 64)).astype(np.float32)
    assert np.array_equal(ffi.avgpool_full(x), ora.avgpool_full(x))


def _oracle_roi_align_levels(feats, scales, rois, counts, PH, f16=False):
    """oracle LevelMapper + RoIAlign (sampling 2) over the FPN levels -> [N, K, PH, PH, C], rows past count zero"""
    N, K = rois.shape[:2]
    out = np.zeros((N, K, PH, PH, feats[0].shape[3]), np.float16 if f16 else np.float32)
    for n in range(N):
        k = counts[n]
        lv = ora.level_map(rois[n, :k])
        for L in range(2, 6):
            idx = np.nonzero(lv == L)[0]
            if len(idx):
                r5 = np.concatenate([np.full((len(idx), 1), n, np.float32), rois[n, idx]], 1)
                out[n, idx] = ora.roi_align(feats[L - 2].astype(np.float32), r5, scales[L - 2], PH, PH, 2).astype(out.dtype)
    return out


@pytest.mark.parametrize("K", [1, 37, 150, 1000, 2048])
def test_roi_prep_order_is_a_level_then_morton_permutation(ffi, K):
    """isegmi_op_roi_prep's launch order: every image's rows exactly once, valid rows first, levels ascending, Morton codes of the centres ascending inside a
    level; its table's last entry names the oracle's level and that level's map size"""
    rng = np.random.default_rng(K)
    N = 3
    scales = [0.25, 0.125, 0.0625, 0.03125]
    shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
    rois = np.stack([_boxes(rng, K, 1344, 800) for _ in range(N)])
    counts = np.array([K, K // 2, 0], np.int32)
    order, tab = ffi.roi_prep(rois, counts, shapes, scales, 256, 7, 7)
    assert order.shape == (N, K) and tab.shape == (N * K, 29, 4)
    tab = tab.reshape(N, K, 29, 4)

    def spread(v):
        return sum(((int(v) >> i) & 1) << (2 * i) for i in range(9))

    for n in range(N):
        ks = order[n] - n * K
        assert sorted(ks.tolist()) == list(range(K))
        c = counts[n]
        assert (ks[:c] < c).all() and list(ks[c:]) == list(range(c, K))   # invalid rows last, in index order
        assert (tab[n, c:] == -1).all()                                    # ... and without a table row
        lv = ora.level_map(rois[n, :c]) if c else np.zeros(0, np.int64)
        assert np.array_equal(tab[n, :c, 28, 0], lv - 2)
        assert all(tuple(tab[n, k, 28, 1:3]) == shapes[lv[k] - 2] for k in range(c))
        keys = []
        for k in ks[:c]:
            b, s = rois[n, k], np.float32(scales[lv[k] - 2])
            cx = int(min(max(np.float32(np.float32(b[0] + b[2]) * np.float32(0.5)) * s, 0), 511))
            cy = int(min(max(np.float32(np.float32(b[1] + b[3]) * np.float32(0.5)) * s, 0), 511))
            keys.append((int(lv[k]), (spread(cy) << 1) | spread(cx), int(k)))
        assert keys == sorted(keys)


@pytest.mark.parametrize("f16", [False, True])
@pytest.mark.parametrize("Cc", [256, 128, 64])
def test_roi_align_from_table_is_the_plain_op_bit_for_bit(ffi, Cc, f16):
    """The FPN heads' launch form (roi_prep's table; one 128-byte channel slice of one RoI per workgroup): against the oracle and against the plain op,
    under roi_prep's order, no order, and a random permutation -- the order is a scheduling hint only."""
    rng = np.random.default_rng(31 + Cc)
    N, K = 2, 203
    shapes = [(50, 84), (25, 42), (13, 21), (7, 11)]
    dt = np.float16 if f16 else np.float32
    feats = [rng.standard_normal((N, h, w, Cc)).astype(dt) for h, w in shapes]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    rois = np.stack([_boxes(rng, K, 336, 200) for _ in range(N)])
    rois[0, 0] = [-50, -40, -10, -5]        # fully outside: every sample invalid
    rois[0, 1] = [10, 10, 10.2, 10.1]       # tiny -> roi size clamp to 1
    rois[1, 2] = [0, 0, 335, 199]
    rois[0, 3] = [300, 150, 420, 260]       # hangs over the right / bottom edge: clamped and invalid samples mixed
    rois[0, 4] = [-30, -20, 40, 50]
    counts = np.array([K, 97], np.int32)
    for PH in (7, 14):
        order, tab = ffi.roi_prep(rois, counts, shapes, scales, Cc, PH, PH, f16=f16)
        ref = _oracle_roi_align_levels(feats, scales, rois, counts, PH, f16)
        plain = (ffi.roi_align_f16(feats, scales, rois, counts, PH, PH) if f16 else ffi.roi_align(feats, scales, rois, counts, PH, PH)[0])
        assert np.array_equal(plain.reshape(ref.shape), ref)
        for o in (order, None, rng.permutation(N * K).astype(np.int32).reshape(N, K)):
            got = ffi.roi_align_ordered(feats, scales, rois, counts, PH, PH, o, tab, f16=f16).reshape(ref.shape)
            assert np.array_equal(got, ref), (PH, Cc, f16)
    # an order that is not a permutation: rows it names are right, rows it leaves out keep what was there, entries outside [0, N*K) are skipped
    order, tab = ffi.roi_prep(rois, counts, shapes, scales, Cc, 7, 7, f16=f16)
    bad = order.copy().reshape(-1)
    left = {int(bad[5]), int(bad[9])}
    bad[5], bad[9] = -1, N * K + 3
    got = ffi.roi_align_ordered(feats, scales, rois, counts, 7, 7, bad.reshape(N, K), tab, f16=f16).reshape(N * K, 7, 7, Cc)
    ref = _oracle_roi_align_levels(feats, scales, rois, counts, 7, f16).reshape(N * K, 7, 7, Cc)
    for r in range(N * K):
        if r in left:
            assert np.isnan(got[r].astype(np.float32)).all()
        else:
            assert np.array_equal(got[r], ref[r])


@pytest.mark.parametrize("f16", [False, True])
def test_roi_align_from_a_garbage_table_stays_inside_the_maps(ffi, f16):
    """The table is caller-supplied device memory: its offsets go through range-checked buffer loads and its level index through a select over the launch's own
    levels, so a table of random words yields finite garbage, not a fault -- and the next call with the real table is right."""
    rng = np.random.default_rng(5)
    N, K, Cc = 1, 64, 64
    shapes = [(20, 30), (10, 15), (5, 8), (3, 4)]
    dt = np.float16 if f16 else np.float32
    feats = [rng.uniform(-1, 1, (N, h, w, Cc)).astype(dt) for h, w in shapes]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    rois = np.stack([_boxes(rng, K, 120, 80) for _ in range(N)])
    counts = np.array([K], np.int32)
    order, tab = ffi.roi_prep(rois, counts, shapes, scales, Cc, 7, 7, f16=f16)
    junk = rng.integers(-2**31, 2**31 - 1, tab.shape, dtype=np.int64).astype(np.int32)
    junk[..., 2:] = tab[..., 2:]   # keep the weights (finite): only the offsets and the level are wild
    got = ffi.roi_align_ordered(feats, scales, rois, counts, 7, 7, order, junk, f16=f16)
    assert np.isfinite(got.astype(np.float32)).all() and np.abs(got.astype(np.float32)).max() <= 1.0 + 1e-3   # convex combinations of map values or zeros
    ref = _oracle_roi_align_levels(feats, scales, rois, counts, 7, f16)
    assert np.array_equal(ffi.roi_align_ordered(feats, scales, rois, counts, 7, 7, order, tab, f16=f16).reshape(ref.shape), ref)


@pytest.mark.parametrize("f16", [False, True])
def test_roi_align_from_table_on_degenerate_boxes_equals_the_plain_op(ffi, f16):
    """Boxes no detector should emit -- reversed corners, zero area, far outside, 1e8-sized, inf, NaN -- and the proposal cap (K = 2048): the table-driven
    launch does exactly what the one-workgroup-per-RoI launch does (NaNs included), and both return."""
    rng = np.random.default_rng(9)
    N, K, Cc = 2, 2048, 64
    shapes = [(40, 60), (20, 30), (10, 15), (5, 8)]
    dt = np.float16 if f16 else np.float32
    feats = [rng.standard_normal((N, h, w, Cc)).astype(dt) for h, w in shapes]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    rois = np.stack([_boxes(rng, K, 240, 160) for _ in range(N)])
    rois[0, 0] = [100, 80, 20, 10]                      # reversed corners
    rois[0, 1] = [50, 50, 50, 50]                       # zero area
    rois[0, 2] = [1e6, 1e6, 1e6 + 30, 1e6 + 30]         # far outside
    rois[0, 3] = [-1e8, -1e8, 1e8, 1e8]                 # covers everything many times over
    rois[0, 4] = [0, 0, np.inf, 40]
    rois[0, 5] = [np.nan, 10, 40, 50]
    rois[0, 6] = [-np.inf, -np.inf, np.inf, np.inf]
    rois[1, 7] = [239.5, 159.5, 239.9, 159.9]           # the last pixel's corner
    counts = np.array([K, 1500], np.int32)
    for PH in (7, 14):
        plain = ffi.roi_align_f16(feats, scales, rois, counts, PH, PH) if f16 else ffi.roi_align(feats, scales, rois, counts, PH, PH)[0]
        order, tab = ffi.roi_prep(rois, counts, shapes, scales, Cc, PH, PH, f16=f16)
        assert sorted(order.reshape(-1).tolist()) == list(range(N * K))
        got = ffi.roi_align_ordered(feats, scales, rois, counts, PH, PH, order, tab, f16=f16)
        assert np.array_equal(got.astype(np.float32), plain.astype(np.float32), equal_nan=True), PH
        assert np.isfinite(plain.reshape(N, K, -1)[1, :1500].astype(np.float32)).all()   # (the wild rows are all in image 0)


def test_roi_align_from_table_rejects_what_it_does_not_cover(ffi):
    from isegmi import _ffi
    rois = np.zeros((1, 4, 4), np.float32)
    cnt = np.array([4], np.int32)
    _, tab = ffi.roi_prep(rois, cnt, [(8, 8)], [0.25], 48, 7, 7)
    with pytest.raises(_ffi.IsegmiError):
        ffi.roi_align_ordered([np.zeros((1, 8, 8, 48), np.float32)], [0.25], rois, cnt, 7, 7, None, tab)             # C = 48
    _, tab = ffi.roi_prep(rois, cnt, [(8, 8)], [0.25], 64, 5, 5)
    with pytest.raises(_ffi.IsegmiError):
        ffi.roi_align_ordered([np.zeros((1, 8, 8, 64), np.float32)], [0.25], rois, cnt, 5, 5, None, tab)             # 5 x 5 bins
    with pytest.raises(_ffi.IsegmiError):
        ffi.roi_prep(np.zeros((1, 2049, 4), np.float32), np.array([2049], np.int32), [(8, 8)], [0.25], 64, 7, 7)    # K > 2048
    with pytest.raises(_ffi.IsegmiError):
        ffi.roi_prep(rois, cnt, [(40000, 40000)], [0.25], 256, 7, 7)                                                  # byte offsets would not fit


@pytest.mark.parametrize("Cc", [256, 64, 36])  # 256 / 64: 8-channel-per-lane kernel; 36: generic 4-channel kernel
def test_roi_align_f16_matches_oracle_on_fp16_features(ffi, Cc):
    """fp16-storage RoIAlign: same fp32 arithmetic on fp16-rounded features, result rounded to fp16 -> exact match."""
    rng = np.random.default_rng(12)
    N, K = 2, 60
    shapes = [(50, 84), (25, 42), (13, 21), (7, 11)]
    feats = [rng.standard_normal((N, h, w, Cc)).astype(np.float16) for h, w in shapes]
    scales = [0.25, 0.125, 0.0625, 0.03125]
    rois = np.stack([_boxes(rng, K, 336, 200, False) for _ in range(N)])
    rois[0, 0] = [-50, -40, -10, -5]
    rois[0, 1] = [10, 10, 10.2, 10.1]
    counts = np.array([K, 41], np.int32)
    for PH in (7, 14):
        out = ffi.roi_align_f16(feats, scales, rois, counts, PH, PH).reshape(N, K, PH, PH, Cc)
        for n in range(N):
            k = counts[n]
            ref_lv = ora.level_map(rois[n, :k])
            for L in range(2, 6):
                idx = np.nonzero(ref_lv == L)[0]
                if len(idx) == 0:
                    continue
                r5 = np.concatenate([np.full((len(idx), 1), n, np.float32), rois[n, idx]], 1)
                ref = ora.roi_align(feats[L - 2].astype(np.float32), r5, scales[L - 2], PH, PH, 2).astype(np.float16)
                assert np.array_equal(out[n, idx], ref), (PH, n, L)
            assert not out[n, k:].any()


@pytest.mark.parametrize("chip_wide", [True, False])
def test_rpn_level_matches_oracle(ffi, chip_wide):
    """chip_wide: the two-kernel NMS (suppression matrix over many CUs + bit scan) against the single-block one; both must
    reproduce the oracle's greedy NMS exactly, including min-size removals, the post_nms cut and ragged counts."""
    from isegmi.maskrcnn import generate_anchors, grid_anchors
    rng = np.random.default_rng(5)
    N, H, W, A = 2, 40, 56, 3
    head = np.concatenate([rng.normal(-2, 2, (N, H, W, A)), rng.normal(0, 0.3, (N, H, W, 4 * A))], -1).astype(np.float32)
    head[0, :4, :4, :A] = 1.25  # ties in the top-k
    head[1, 10:30, 10:50, A:] *= 0.02  # a region of near-identical boxes: long suppression chains inside 64-box chunks
    anchors = grid_anchors(H, W, 8, generate_anchors(8, 64, (0.5, 1.0, 2.0)))
    hw = np.array([[300, 440], [320, 448]], np.int32)
    for pre, post, min_size in ((1000, 1000, 0.0), (300, 50, 0.0), (1000, 37, 0.0), (1024, 1000, 40.0), (70, 70, 0.0)):
        got = ffi.rpn_level(head, anchors, hw, A, pre, post, min_size=min_size, chip_wide=chip_wide)
        for n in range(N):
            rb, rs = ora.rpn_level(head[n, ..., :A].reshape(-1), head[n, ..., A:].reshape(-1, 4), anchors, pre, post, 0.7, min_size,
                                   float(hw[n, 1]), float(hw[n, 0]))
            assert np.array_equal(got[n][1], rs) and np.array_equal(got[n][0], rb), (pre, post, min_size, n)
            assert len(rs) > 10


@pytest.mark.parametrize("N", [1, 2, 3])
def test_rpn_levels_batched_matches_oracle_and_per_level(ffi, N):
    """the RPN selection with (level, image) as ONE batch dimension (isegmi_op_rpn_levels: five launches for all levels) against the oracle's
    per-level selection and against the per-level op, level by level -- ties in the top-k, a level with fewer anchors than pre_nms (P6: 6 x 8 x 3 = 144),
    levels above and below the 12 288-key slice of the two-level top-k, min-size removals and the post_nms cut included."""
    from isegmi.maskrcnn import generate_anchors, grid_anchors
    rng = np.random.default_rng(50 + N)
    A = 3
    shapes = [(88, 120, 4, 32), (44, 60, 8, 64), (22, 30, 16, 128), (11, 15, 32, 256), (6, 8, 64, 512)]   # (H, W, stride, anchor size): 31 680 ... 144 anchors
    heads, anchors = [], []
    for li, (H, W, stride, size) in enumerate(shapes):
        h = np.concatenate([rng.normal(-2, 2, (N, H, W, A)), rng.normal(0, 0.3, (N, H, W, 4 * A))], -1).astype(np.float32)
        h[0, :3, :3, :A] = 1.25                       # ties in the top-k
        if li == 1:
            h[N - 1, 10:30, 10:50, A:] *= 0.02        # near-identical boxes: long suppression chains
        heads.append(h); anchors.append(grid_anchors(H, W, stride, generate_anchors(stride, size, (0.5, 1.0, 2.0))))
    hw = np.array([[340 + 7 * n, 470 - 5 * n] for n in range(N)], np.int32)
    for pre, post, min_size in ((1000, 1000, 0.0), (300, 50, 0.0), (1024, 1000, 24.0), (1000, 37, 0.0)):
        got = ffi.rpn_levels(heads, anchors, hw, A, pre, post, min_size=min_size)
        for l in range(len(shapes)):
            one = ffi.rpn_level(heads[l], anchors[l], hw, A, pre, post, min_size=min_size)
            for n in range(N):
                rb, rs = ora.rpn_level(heads[l][n, ..., :A].reshape(-1), heads[l][n, ..., A:].reshape(-1, 4), anchors[l], pre, post, 0.7, min_size,
                                       float(hw[n, 1]), float(hw[n, 0]))
                assert np.array_equal(got[l][n][1], rs) and np.array_equal(got[l][n][0], rb), (pre, post, min_size, l, n)
                assert np.array_equal(got[l][n][1], one[n][1]) and np.array_equal(got[l][n][0], one[n][0])
            assert sum(len(got[l][n][1]) for n in range(N)) > 5


def test_rpn_levels_batched_at_the_bench_size(ffi):
    """the five levels of the 800 x 1344 canvas (201 600 ... 819 anchors per image: P2 is cut into 17 slices, P6 holds fewer keys than pre_nms), two images;
    checked against the per-level op (itself checked against the oracle above and, end to end, by the full-size Mask R-CNN test)"""
    from isegmi.maskrcnn import generate_anchors, grid_anchors
    rng = np.random.default_rng(77)
    A, N = 3, 2
    shapes = [(200, 336, 4, 32), (100, 168, 8, 64), (50, 84, 16, 128), (25, 42, 32, 256), (13, 21, 64, 512)]
    heads = [np.concatenate([rng.normal(-2, 2, (N, H, W, A)), rng.normal(0, 0.3, (N, H, W, 4 * A))], -1).astype(np.float32) for H, W, _, _ in shapes]
    anchors = [grid_anchors(H, W, s, generate_anchors(s, z, (0.5, 1.0, 2.0))) for H, W, s, z in shapes]
    hw = np.array([[800, 1333], [768, 1344]], np.int32)
    got = ffi.rpn_levels(heads, anchors, hw, A, 1000, 1000)
    for l in range(5):
        one = ffi.rpn_level(heads[l], anchors[l], hw, A, 1000, 1000)
        for n in range(N):
            assert np.array_equal(got[l][n][1], one[n][1]) and np.array_equal(got[l][n][0], one[n][0]), (l, n)
            assert len(one[n][1]) > 100


def test_rpn_single_map_6000_matches_oracle(ffi):
    """R-50-C4 style RPN: one stride-16 map, 15 anchors per location (5 sizes x 3 ratios), PRE_NMS_TOP_N_TEST 6000 ->
    NMS 0.7 -> 1000 (README.md:267-269).  Exercises the k <= 8192 top-k and the 6144-box NMS kernels."""
    from isegmi.maskrcnn import generate_anchors_multi, grid_anchors
    rng = np.random.default_rng(15)
    N, H, W, A = 2, 40, 60, 15
    head = np.concatenate([rng.normal(-2, 2, (N, H, W, A)), rng.normal(0, 0.3, (N, H, W, 4 * A))], -1).astype(np.float32)
    head[1, :6, :6, :A] = 0.75  # ties in the top-k
    anchors = grid_anchors(H, W, 16, generate_anchors_multi(16, (32, 64, 128, 256, 512), (0.5, 1.0, 2.0)))
    assert anchors.shape == (H * W * A, 4)
    hw = np.array([[640, 960], [600, 900]], np.int32)
    for pre, post in ((6000, 1000), (3000, 300)):
        got = ffi.rpn_level(head, anchors, hw, A, pre, post)
        for n in range(N):
            rb, rs = ora.rpn_level(head[n, ..., :A].reshape(-1), head[n, ..., A:].reshape(-1, 4), anchors, pre, post, 0.7, 0.0,
                                   float(hw[n, 1]), float(hw[n, 0]))
            assert np.array_equal(got[n][1], rs) and np.array_equal(got[n][0], rb), (pre, n)
            assert len(rs) > 100


@pytest.mark.parametrize("chip_wide", [True, False])
def test_box_postprocess_matches_oracle(ffi, chip_wide):
    """chip_wide: classes with more than 128 candidates handed to the chip-wide suppression matrix (three launches) or kept in their own block"""
    import functools
    ffi_box = functools.partial(ffi.box_postprocess, chip_wide=chip_wide)
    rng = np.random.default_rng(9)
    N, R, ncls = 2, 1000, 81
    logits = rng.normal(0, 1.0, (N, R, ncls)).astype(np.float32)
    logits[..., 0] += 2.0
    logits[..., [7, 31, 56]] += 2.5
    regr = rng.normal(0, 0.5, (N, R, 4 * ncls)).astype(np.float32)
    props = np.stack([_boxes(rng, R) for _ in range(N)])
    cnt = np.array([R, 613], np.int32)
    hw = np.array([[800, 1333], [750, 1200]], np.int32)
    got = ffi_box(logits, regr, props, cnt, hw)
    for n in range(N):
        k = cnt[n]
        rb, rs, rl = ora.box_postprocess(logits[n, :k], regr[n, :k], props[n, :k], float(hw[n, 1]), float(hw[n, 0]), cap=100)
        assert len(rs) == 100
        assert np.array_equal(got[n][2], rl) and np.array_equal(got[n][1], rs) and np.array_equal(got[n][0], rb)
    # crowded classes (several hundred candidates each: the in-block bitmask NMS) on tightly clustered, near-identical proposals
    props3 = props.copy()
    props3[:, 200:] = props3[:, :800] + rng.normal(0, 1.5, (N, 800, 4)).astype(np.float32)
    logits3 = logits.copy(); logits3[..., [7, 31]] += 1.5
    got = ffi_box(logits3, regr * 0.05, props3, cnt, hw)
    for n in range(N):
        k = cnt[n]
        rb, rs, rl = ora.box_postprocess(logits3[n, :k], regr[n, :k] * 0.05, props3[n, :k], float(hw[n, 1]), float(hw[n, 0]), cap=100)
        assert np.array_equal(got[n][2], rl) and np.array_equal(got[n][1], rs) and np.array_equal(got[n][0], rb)
    # exact ties at the kth value: upstream's `cls_scores >= kthvalue` keeps ALL of them, so more than det_per_img detections come back
    # when the buffers have the rows for them (cap > det_per_img); with the default capacity the list is cut at det_per_img in output order
    Rt = 300
    gx, gy = np.meshgrid(np.arange(20) * 60.0, np.arange(15) * 50.0)
    propt = np.stack([gx.ravel(), gy.ravel(), gx.ravel() + 40, gy.ravel() + 30], 1).astype(np.float32)[None]   # disjoint boxes: NMS keeps all
    logt = np.full((1, Rt, ncls), -4.0, np.float32)
    logt[0, :60, 5] = 3.0; logt[0, 60:210, 9] = 2.0      # 60 high scores of class 5, then 150 IDENTICAL scores of class 9: 210 survivors
    regt = np.zeros((1, Rt, 4 * ncls), np.float32)
    cntt, hwt = np.array([Rt], np.int32), np.array([[800, 1333]], np.int32)
    for cap, want in ((100, 100), (128, 128), (256, 210)):
        (gb, gs, gl), = ffi_box(logt, regt, propt, cntt, hwt, cap=cap)
        rb, rs, rl = ora.box_postprocess(logt[0], regt[0], propt[0], 1333.0, 800.0, cap=cap)
        assert len(rs) == want == len(gs), (cap, len(rs), len(gs))
        assert np.array_equal(gl, rl) and np.array_equal(gs, rs) and np.array_equal(gb, rb)
    assert (gl == 5).sum() == 60 and (gl == 9).sum() == 150 and len(np.unique(gs[gl == 9])) == 1
    # fewer than det_per_img survivors: everything kept, order preserved
    logits2 = logits.copy(); logits2[..., 0] += 6.0
    got = ffi_box(logits2, regr, props, cnt, hw)
    for n in range(N):
        k = cnt[n]
        rb, rs, rl = ora.box_postprocess(logits2[n, :k], regr[n, :k], props[n, :k], float(hw[n, 1]), float(hw[n, 0]), cap=100)
        assert 0 < len(rs) < 100
        assert np.array_equal(got[n][2], rl) and np.array_equal(got[n][1], rs) and np.array_equal(got[n][0], rb)


def test_mask_tail_and_paste_match_oracle(ffi):
    rng = np.random.default_rng(13)
    R, HW, Cc = 37, 784, 256
    feat = np.maximum(rng.standard_normal((R, HW, Cc)), 0).astype(np.float32)
    w = (rng.standard_normal((81, Cc)) * 0.1).astype(np.float32); b = rng.standard_normal(81).astype(np.float32)
    labels = rng.integers(1, 81, R).astype(np.int32)
    got = ffi.mask_logits_select(feat, w, b, labels)
    assert np.array_equal(got, ora.mask_logits_select(feat, w, b, labels))
    N, K, im_h, im_w = 2, 20, 203, 317
    masks = rng.uniform(0, 1, (N, K, 28, 28)).astype(np.float32)
    boxes = np.stack([_boxes(rng, K, im_w, im_h, False) for _ in range(N)])
    boxes[0, 0] = [-30.5, -20.2, 40.7, 60.1]; boxes[0, 1] = [250.3, 150.9, 400.0, 300.0]; boxes[0, 2] = [10, 10, 10.4, 10.2]
    cnt = np.array([K, 11], np.int32)
    out = ffi.paste_masks(masks, boxes, cnt, im_h, im_w)
    for n in range(N):
        ref = ora.paste_masks(masks[n, : cnt[n]], boxes[n, : cnt[n]], im_h, im_w)
        assert np.array_equal(out[n, : cnt[n]], ref)
        assert not out[n, cnt[n]:].any()
