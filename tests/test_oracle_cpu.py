"""CPU tests of the oracle: hand-derived known answers, cross-checks against torch-CPU /
numpy where an independent implementation exists, and the committed golden vectors (drift guard)."""
import os

import numpy as np
import pytest

from oracle import ora

G = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    return np.load(os.path.join(G, name + ".npz"))


def test_detmath_accuracy():
    x = np.linspace(-30, 30, 20001).astype(np.float32)
    x64 = x.astype(np.float64)
    assert np.max(np.abs(ora.map_f32(x, 0) / np.exp(x64) - 1)) < 2.5e-7
    assert np.max(np.abs(ora.map_f32(x, 1) - 1 / (1 + np.exp(-x64)))) < 1.5e-7
    assert np.max(np.abs(ora.map_f32(x, 2) - np.tanh(x64))) < 2e-7
    xp = np.abs(x) + 1e-4
    assert np.max(np.abs(ora.map_f32(xp, 3) - np.log2(xp.astype(np.float64)))) < 1e-6
    assert ora.map_f32(np.array([-200.0], np.float32), 0)[0] == 0.0 and ora.map_f32(np.array([0.0], np.float32), 0)[0] == 1.0


def test_conv_vs_torch_and_fma_chain():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 10, 13, 32)).astype(np.float32); w = (rng.standard_normal((24, 3, 3, 32)) * 0.1).astype(np.float32)
    for stride, pad in ((1, 1), (2, 1), (1, 0)):
        y = ora.conv2d(x, w, stride, pad)
        t = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(w).permute(0, 3, 1, 2).double(),
                                       stride=stride, padding=pad).permute(0, 2, 3, 1).numpy()
        assert np.max(np.abs(y - t)) < 2e-5
    # the documented rounding sequence: k-ordered fmaf chain over (r, s, c) from +0
    y = ora.conv2d(x, w, 1, 1)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    import math
    for (n, ho, wo, co) in ((0, 0, 0, 0), (1, 4, 7, 13), (1, 9, 12, 23)):
        acc = np.float32(0)
        for r in range(3):
            for s in range(3):
                for c in range(32):
                    a, b = float(xp[n, ho + r, wo + s, c]), float(w[co, r, s, c])
                    acc = np.float32(math.fma(a, b, float(acc))) if hasattr(math, "fma") else np.float32(np.float64(a) * np.float64(b) + np.float64(acc))
        assert y[n, ho, wo, co] == acc
    # more than 128 input channels: groups of 128 channels outermost, then (r, s), then the group's channels (round 3; ora_conv2d's header)
    x = rng.standard_normal((1, 5, 6, 320)).astype(np.float32); w = (rng.standard_normal((8, 3, 3, 320)) * 0.05).astype(np.float32)
    y = ora.conv2d(x, w, 1, 1)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    differs = 0
    for (ho, wo, co) in ((0, 0, 0), (2, 3, 5), (4, 5, 7), (1, 1, 2), (3, 2, 6)):
        acc, plain = np.float32(0), np.float32(0)
        fma = lambda a, b, c: np.float32(math.fma(a, b, float(c))) if hasattr(math, "fma") else np.float32(np.float64(a) * np.float64(b) + np.float64(c))
        for cg in range(0, 320, 128):
            for r in range(3):
                for s in range(3):
                    for c in range(cg, min(cg + 128, 320)):
                        acc = fma(float(xp[0, ho + r, wo + s, c]), float(w[co, r, s, c]), acc)
        for r in range(3):
            for s in range(3):
                for c in range(320):
                    plain = fma(float(xp[0, ho + r, wo + s, c]), float(w[co, r, s, c]), plain)
        assert y[0, ho, wo, co] == acc
        differs += int(plain != acc)
    assert differs > 0   # the two orders are different roundings of the same sum: the test would not notice the order otherwise
    # a 1x1 convolution has one tap: the grouped order IS the plain one
    w1 = (rng.standard_normal((4, 1, 1, 320)) * 0.05).astype(np.float32)
    y1 = ora.conv2d(x, w1, 1, 0)
    acc = np.float32(0)
    for c in range(320):
        acc = fma(float(x[0, 2, 3, c]), float(w1[1, 0, 0, c]), acc)
    assert y1[0, 2, 3, 1] == acc


def test_pool_resize_deconv_vs_torch():
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    rng = np.random.default_rng(1)
    a = rng.standard_normal((2, 19, 23, 8)).astype(np.float32)
    t = torch.from_numpy(a).permute(0, 3, 1, 2)
    assert np.array_equal(ora.maxpool(a, 3, 2, 1), F.max_pool2d(t, 3, 2, 1).permute(0, 2, 3, 1).numpy())
    assert np.array_equal(ora.maxpool(a, 1, 2, 0), a[:, ::2, ::2])
    y = ora.resize_bilinear(a, 35, 41)
    assert np.max(np.abs(y - F.interpolate(t, size=(35, 41), mode="bilinear", align_corners=False).permute(0, 2, 3, 1).numpy())) < 1e-5
    lat = rng.standard_normal((2, 38, 46, 8)).astype(np.float32)
    assert np.array_equal(ora.upsample_nearest2x_add(a, lat), lat + np.repeat(np.repeat(a, 2, 1), 2, 2))
    wd = rng.standard_normal((8, 6, 2, 2)).astype(np.float32); b = rng.standard_normal(6).astype(np.float32)
    ref = torch.relu(F.conv_transpose2d(t, torch.from_numpy(wd), torch.from_numpy(b), stride=2)).permute(0, 2, 3, 1).numpy()
    assert np.max(np.abs(ora.deconv2x2(a, wd, b, 1) - ref)) < 1e-5


def test_conv_leaky_activations_known_answers():
    """act 3 = LeakyReLU(0.1) after the (optional) residual; act 4 = LeakyReLU(0.1) FIRST, then the residual (DarkNetBlock)."""
    x = np.zeros((1, 1, 2, 32), np.float32); x[0, 0, 0, 0] = 2.0; x[0, 0, 1, 0] = -3.0
    w = np.zeros((1, 1, 1, 32), np.float32); w[0, 0, 0, 0] = 1.0
    res = np.full((1, 1, 2, 1), 10.0, np.float32)
    assert np.array_equal(ora.conv2d(x, w, 1, 0, None, None, None, 3).ravel(), np.float32([2.0, np.float32(-3.0) * np.float32(0.1)]))
    assert np.array_equal(ora.conv2d(x, w, 1, 0, None, None, res, 4).ravel(), np.float32([12.0, np.float32(-3.0) * np.float32(0.1) + np.float32(10.0)]))
    assert np.array_equal(ora.conv2d(x, w, 1, 0, None, None, -res, 3).ravel(), np.float32([np.float32(-8.0) * np.float32(0.1), np.float32(-13.0) * np.float32(0.1)]))


def test_deform_im2col_known_answers():
    """DCNv2 sampling stage (YOLACT++ backbones): zero offsets + saturated mask == plain zero-padded im2col, exactly; a
    half-pixel shift averages neighbours; an independent float64 restatement agrees on random offsets incl. out-of-range."""
    rng = np.random.default_rng(3)
    N, H, W, C = 2, 7, 9, 8
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    for stride in (1, 2):
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        om = np.zeros((N, Ho, Wo, 27), np.float32); om[..., 18:] = 30.0
        col = ora.deform_im2col(x, om, 3, 3, stride, 1, 1).reshape(N, Ho, Wo, 9, C)
        xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
        for i in range(3):
            for j in range(3):
                ref = xp[:, i:i + (Ho - 1) * stride + 1:stride, j:j + (Wo - 1) * stride + 1:stride]
                assert np.array_equal(col[:, :, :, i * 3 + j], ref), (stride, i, j)
    om = np.zeros((N, H, W, 27), np.float32); om[..., 18:] = 30.0
    om[..., 9] = 0.5  # centre tap (k = 4): dx = +0.5 (channel 2k+1)
    col = ora.deform_im2col(x, om).reshape(N, H, W, 9, C)
    ref = 0.5 * x[:, :, :-1] + 0.5 * x[:, :, 1:]
    assert np.array_equal(col[:, :, :-1, 4], ref)
    assert np.array_equal(col[:, :, -1, 4], 0.5 * x[:, :, -1])  # right neighbour outside the image contributes 0
    # random offsets / masks against float64
    om = rng.normal(0, 2.5, (N, H, W, 27)).astype(np.float32)
    col = ora.deform_im2col(x, om).reshape(N, H, W, 9, C)
    ref = np.zeros((N, H, W, 9, C))
    for n in range(N):
        for ho in range(H):
            for wo in range(W):
                for k in range(9):
                    h = ho - 1 + k // 3 + float(om[n, ho, wo, 2 * k]); w = wo - 1 + k % 3 + float(om[n, ho, wo, 2 * k + 1])
                    if not (h > -1 and w > -1 and h < H and w < W):
                        continue
                    hl, wl = int(np.floor(h)), int(np.floor(w))
                    lh, lw = h - hl, w - wl
                    v = np.zeros(C)
                    for (yy, xx, ww) in ((hl, wl, (1 - lh) * (1 - lw)), (hl, wl + 1, (1 - lh) * lw), (hl + 1, wl, lh * (1 - lw)), (hl + 1, wl + 1, lh * lw)):
                        if 0 <= yy < H and 0 <= xx < W:
                            v += ww * x[n, yy, xx].astype(np.float64)
                    ref[n, ho, wo, k] = v / (1.0 + np.exp(-float(om[n, ho, wo, 18 + k])))
    assert np.allclose(col, ref, rtol=1e-5, atol=1e-6)
    assert (ref == 0).reshape(-1, C).all(1).sum() > 20  # the out-of-range branch was exercised


def test_topk_total_order():
    s = np.array([0.5, 0.9, 0.5, 0.1, 0.9, 0.5], np.float32)
    v, i = ora.topk(s, 4)
    assert list(i) == [1, 4, 0, 2] and list(v) == [np.float32(0.9)] * 2 + [np.float32(0.5)] * 2
    v, i = ora.topk(s, 10)
    assert list(i) == [1, 4, 0, 2, 5, 3]


def test_nms_known_answers():
    b = np.array([[0, 0, 9, 9], [0, 0, 9, 9], [0, 0, 9, 4], [100, 100, 120, 120]], np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    assert list(ora.nms(b, s, 0.5, 1, 0)) == [0, 2, 3]   # IoU(0,2) = 50/100 exactly = thr: '>' keeps it
    assert list(ora.nms(b, s, 0.5, 1, 1)) == [0, 3]      # '>=' suppresses it
    chain = np.array([[0, 0, 99, 99], [0, 0, 99, 59], [0, 0, 99, 35]], np.float32)
    assert list(ora.nms(chain, np.array([0.9, 0.8, 0.7], np.float32), 0.5, 1, 0)) == [0, 2]  # suppressed B must not suppress C
    assert list(ora.nms(b, s[::-1].copy(), 0.5, 1, 0)) == [3, 2, 1]  # visiting order follows the scores
    assert list(ora.nms(b, s, 0.5, 1, 0, max_keep=2)) == [0, 2]


def test_box_decode_known_answer():
    a = np.array([[0, 0, 9, 19]], np.float32)  # w=10 h=20 ctr (5,10)
    d = np.array([[0.1, -0.1, 0.0, np.log(2.0)]], np.float32)
    o = ora.decode_boxes(a, d, (1, 1, 1, 1), 1000, 1000, clip=False)[0]
    assert np.allclose(o, [6 - 5, 8 - 20, 6 + 5 - 1, 8 + 20 - 1], atol=1e-4)
    big = ora.decode_boxes(a, np.array([[0, 0, 50.0, 50.0]], np.float32), (1, 1, 1, 1), 100, 100, clip=True)[0]
    assert np.array_equal(big, [0, 0, 99, 99])  # dw clamp log(1000/16) then clip to (w-1,h-1)


def test_roi_align_known_answers():
    const = np.full((1, 20, 30, 3), 2.25, np.float32)
    r = np.array([[0, 4, 6, 30, 28]], np.float32)
    assert np.allclose(ora.roi_align(const, r, 0.5, 7, 7, 2), 2.25, atol=1e-6)
    ramp = np.tile(np.arange(30, dtype=np.float32)[None, None, :, None], (1, 30, 1, 1))
    o = ora.roi_align(ramp, np.array([[0, 8, 8, 24, 24]], np.float32), 1.0, 4, 4, 2)[0, :, :, 0]
    # bin width 4, samples at x = 8 + 4*pw + {1, 3} -> mean = 10 + 4*pw
    assert np.allclose(o, np.tile(10 + 4 * np.arange(4, dtype=np.float32), (4, 1)), atol=1e-5)
    assert not ora.roi_align(ramp, np.array([[0, -50, -40, -10, -5]], np.float32), 1.0, 7, 7, 2).any()
    assert list(ora.level_map(np.array([[0, 0, 223, 223], [0, 0, 111, 111], [0, 0, 10, 10], [0, 0, 1000, 1000]], np.float32))) == [4, 3, 2, 5]


def test_yolact_decode_and_fast_nms_known_answers():
    pri = np.array([[0.5, 0.5, 0.2, 0.2]], np.float32)
    assert np.allclose(ora.yolact_decode(np.zeros((1, 4), np.float32), pri), [[0.4, 0.4, 0.6, 0.6]], atol=1e-7)
    # three priors of class 1: A and B overlap heavily (B dropped), C suppressed only by B in greedy NMS but fast-NMS drops it too
    boxes = np.array([[0, 0, 1, 1], [0, 0, 1, 0.9], [0, 0, 1, 0.5]], np.float32)  # iou(A,B)=.9 iou(A,C)=.5 iou(B,C)=.556
    conf = np.zeros((3, 3), np.float32); conf[:, 1] = [0.9, 0.8, 0.7]; conf[:, 0] = 0.05; conf[:, 2] = 0.01
    d = ora.yolact_detect(conf, boxes, np.zeros((3, 4), np.float32), 0.05, 0.5, 200, 100)
    keep1 = d["prior"][d["cls"] == 0]
    assert list(keep1) == [0]  # C has iou .556 > .5 with the (already suppressed) B: fast-NMS still removes it
    empty = ora.yolact_detect(np.tile(np.array([[0.99, 0.005, 0.005]], np.float32), (3, 1)), boxes, np.zeros((3, 4), np.float32))
    assert len(empty["score"]) == 0


def test_semantic_fork_known_answers():
    """SURVEY 7.2 / App. A.6, A.7 forks of the oracle, each on a hand-derived case where its two sides differ."""
    # --- NMS inside the RPN level: zero deltas return the anchors, so `anchors` are the candidate boxes
    boxes = np.array([[0, 0, 9, 0], [3, 0, 9, 0], [20, 0, 29, 9], [20, 0, 29, 6]], np.float32)   # IoU 7/10 and 70/100 with +1; 0/0 and 54/81 without
    logit = np.log(np.array([0.9, 0.8, 0.7, 0.6]) / (1 - np.array([0.9, 0.8, 0.7, 0.6]))).astype(np.float32)
    z = np.zeros((4, 4), np.float32)
    keep = lambda thr, flags: ora.rpn_level(logit, z, boxes, 4, 4, thr, 0.0, 400.0, 400.0, flags)[0].tolist()
    assert keep(0.7, 0) == boxes.tolist()                       # 0.7 is not > 0.7
    assert keep(0.7, 1) == boxes[[0, 2]].tolist()               # >= suppresses both ties
    assert keep(0.68, 0) == boxes[[0, 2]].tolist()              # +1: both IoUs are 0.7 > 0.68
    assert keep(0.68, 2) == boxes.tolist()                      # plain areas: NaN and 0.667
    assert keep(0.66, 2) == boxes[[0, 1, 2]].tolist()           # 0.667 > 0.66; 0/0 never suppresses
    # --- box post-processing: output order inside a class
    props = np.array([[0, 0, 9, 9], [0, 0, 9, 4], [50, 50, 80, 90], [100, 20, 130, 60]], np.float32)
    logits = np.full((4, 81), -6.0, np.float32); logits[:, 0] = 0.0
    logits[:, 5] = [2.0, 1.0, 3.0, 4.0]
    regr = np.zeros((4, 324), np.float32)
    order = lambda flags: [props.tolist().index(b) for b in ora.box_postprocess(logits, regr, props, 400.0, 400.0, nms_flags=flags)[0].tolist()]
    assert order(0) == [3, 2, 0, 1] and order(4) == [0, 1, 2, 3]          # score order / proposal-index order
    assert order(1) == [3, 2, 0] and order(5) == [0, 2, 3]                # IoU(0, 1) = 50/100 = 0.5: suppressed by >= only
    assert order(3) == [3, 2, 0, 1]                                       # plain areas: 36/81 < 0.5
    # --- RoIAlign aligned: on a ramp f[y, x] = x the bins of RoI [8, 22] average 8 + (pw + 0.5) * 2 (legacy) and 0.5 less (aligned)
    ramp = np.tile(np.arange(40, dtype=np.float32)[None, :, None], (12, 1, 1))[None]
    roi = np.array([[0, 8.0, 2.0, 22.0, 9.0]], np.float32)
    for aligned in (0, 1):
        got = ora.roi_align(ramp, roi, 1.0, 7, 7, 2, aligned)[0, 3, :, 0]
        assert np.allclose(got, 8.0 + (np.arange(7) + 0.5) * 2.0 - 0.5 * aligned, atol=1e-5)
    # a RoI thinner than a pixel: the legacy op widens it to one pixel, aligned keeps 0.25 -> all 14 sample columns inside [10.0, 10.25) - 0.5
    thin = np.array([[0, 10.0, 2.0, 10.25, 9.0]], np.float32)
    leg, ali = ora.roi_align(ramp, thin, 1.0, 7, 7, 2, 0)[0, 0, :, 0], ora.roi_align(ramp, thin, 1.0, 7, 7, 2, 1)[0, 0, :, 0]
    assert leg[-1] - leg[0] > 0.8 and ali[-1] - ali[0] < 0.25 and 9.5 <= ali[0] <= ali[-1] < 9.75
    # --- Yolact fast_nms(second_threshold): a prior that passed the pre-filter on class 3 is also ranked in class 7 with a sub-threshold score
    prob = np.full((4, 81), 1e-4, np.float32)
    prob[0, 3] = 0.6; prob[0, 7] = 0.04; prob[1, 7] = 0.5
    bx = np.array([[0.1, 0.1, 0.2, 0.2], [0.3, 0.3, 0.4, 0.4], [0.5, 0.5, 0.6, 0.6], [0.7, 0.7, 0.8, 0.8]], np.float32)
    mk = np.zeros((4, 32), np.float32)
    pairs = lambda st: sorted(zip(*(ora.yolact_detect(prob, bx, mk, second_threshold=st)[k].tolist() for k in ("prior", "cls"))))
    assert pairs(0) == [(0, 2), (0, 6), (1, 2), (1, 6)] or (0, 6) in pairs(0)
    assert (0, 6) not in pairs(1) and (0, 2) in pairs(1) and (1, 6) in pairs(1)
    # --- FrozenBatchNorm eps
    from oracle.maskrcnn_ref import _frozen_bn
    sd = {"bn.weight": np.array([2.0], np.float32), "bn.bias": np.array([1.0], np.float32), "bn.running_mean": np.array([3.0], np.float32),
          "bn.running_var": np.array([4.0], np.float32)}
    sc, sh = _frozen_bn(sd, "bn"); assert sc[0] == 1.0 and sh[0] == -2.0
    sc, sh = _frozen_bn(sd, "bn", 5.0); assert np.allclose(sc[0], 2.0 / 3.0) and np.allclose(sh[0], 1.0 - 2.0)


def test_conv_split_k_is_the_stated_sum():
    """ora.conv2d(ksplit=4): ((p0 + p1) + p2) + p3 of four k-ordered fmaf chains over equal ranges of the 32-channel chunks in the K order (channel groups of
    128 outermost, then (r, s), then the group's chunks) -- restated here in plain Python for a few outputs; ksplit = 1 is the default chain."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 9, 11, 256)).astype(np.float32)
    w = (rng.standard_normal((64, 3, 3, 256)) * 0.05).astype(np.float32)
    one, four = ora.conv2d(x, w, 1, 1), ora.conv2d(x, w, 1, 1, ksplit=4)
    assert np.array_equal(one, ora.conv2d(x, w, 1, 1, ksplit=1)) and not np.array_equal(one, four)
    assert np.abs(one - four).max() < 1e-4

    def fma32(a, b, c):   # fp32 x fp32 is exact in fp64; the one rounding of the fp64 sum to fp32 can differ from a true fma only by double rounding
        return np.float32(np.float64(a) * np.float64(b) + np.float64(c))
    chunks = [(r, s, c0) for cg in (0, 128) for r in range(3) for s in range(3) for c0 in range(cg, cg + 128, 32)]
    L = -(-len(chunks) // 4)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    for (ho, wo, co) in [(0, 0, 0), (4, 5, 17), (8, 10, 63), (3, 0, 40)]:
        tot = None
        for q in range(4):
            acc = np.float32(0)
            for (r, s, c0) in chunks[q * L:(q + 1) * L]:
                for c in range(c0, c0 + 32):
                    acc = fma32(xp[0, ho + r, wo + s, c], w[co, r, s, c], acc)
            tot = acc if tot is None else np.float32(tot + acc)
        assert tot == four[0, ho, wo, co]
    assert ora.conv_split_qualifies(1225, 256, 3, 3, 256) and not ora.conv_split_qualifies(1225, 1024, 1, 1, 256) and not ora.conv_split_qualifies(38088, 256, 3, 3, 256)


def test_paste_known_answer():
    m = np.full((1, 28, 28), 0.9, np.float32)
    o = ora.paste_masks(m, np.array([[50, 60, 250, 300]], np.float32), 384, 500)[0]
    ys, xs = np.nonzero(o.any(1))[0], np.nonzero(o.any(0))[0]
    assert (ys[0], ys[-1], xs[0], xs[-1]) == (60, 299, 50, 249)  # the 1-px zero border of the padded mask trims the box edge
    assert o.sum() > 0.95 * 200 * 240


def test_box_postprocess_kth_value_cut():
    rng = np.random.default_rng(3)
    R = 300
    logits = rng.normal(0, 1, (R, 81)).astype(np.float32); logits[:, 5] += 4
    regr = np.zeros((R, 324), np.float32)
    c = rng.uniform(50, 900, (R, 2)); props = np.concatenate([c, c + 20], 1).astype(np.float32)  # mostly disjoint boxes
    b, s, l = ora.box_postprocess(logits, regr, props, 1333, 800, cap=128)
    assert len(s) == 100 and np.all(np.diff(l) >= 0)  # class-major order, exactly det_per_img without ties
    b2, s2, l2 = ora.box_postprocess(logits, regr, props, 1333, 800, det_per_img=1000, cap=1000)
    thr = np.sort(s2)[::-1][99]
    assert np.array_equal(s, s2[s2 >= thr])


def test_front_end_known_answers():
    """M1 / Y1 (App. A.0 constants, A.1 bilinear rule) on cases small enough to do by hand"""
    # Y1, in == out: the resize is the identity -> ((float)u8 - mean) / std per BGR channel, then RGB order
    x = np.array([[[[0, 128, 255], [10, 20, 30]]]], np.uint8)            # [1, 1, 2, 3]
    y = ora.fast_base_transform(np.repeat(x, 2, 1), 2)
    m, sd = np.float32(ora.YOLACT_MEANS), np.float32(ora.YOLACT_STD)
    want = ((x[0, 0].astype(np.float32) - m) / sd)[:, ::-1]
    assert np.array_equal(y[0, 0], want) and np.array_equal(y[0, 1], want)
    # Y1, 2 -> 4 along a row of (0, 100): src = (d + .5) / 2 - .5 -> 0 (clamped), .25, .75, 1.25 -> values 0, 25, 75, 100 (the right tap clamps)
    r = np.zeros((1, 1, 2, 3), np.uint8); r[0, 0, 1] = 100
    y = ora.fast_base_transform(np.repeat(r, 2, 1), 4)                   # source 2 x 2 (both rows equal), target 4 x 4
    v = np.array([0, 25, 75, 100], np.float32)
    for c in range(3):
        assert np.array_equal(y[0, 0, :, 2 - c], (v - m[c]) / sd[c]), c
    assert np.array_equal(y[0, 0], y[0, 3])
    # darknet53: x / 255, RGB
    y = ora.fast_base_transform(np.full((1, 3, 3, 3), 255, np.uint8), 3, darknet=True)
    assert np.array_equal(y, np.ones((1, 3, 3, 3), np.float32))
    # M1: minus PIXEL_MEAN (BGR kept), zero padding to the batch's largest size rounded up to 32, unpadded sizes remembered
    a = np.full((3, 40, 3), 200, np.uint8); b = np.full((33, 2, 3), 7, np.uint8)
    out, hw = ora.to_image_list([a, b])
    assert out.shape == (2, 64, 64, 3) and hw.tolist() == [[3, 40], [33, 2]]
    pm = np.float32(ora.PIXEL_MEAN)
    assert np.array_equal(out[0, :3, :40], np.broadcast_to(np.float32(200) - pm, (3, 40, 3))) and not out[0, 3:].any() and not out[0, :, 40:].any()
    assert np.array_equal(out[1, :33, :2], np.broadcast_to(np.float32(7) - pm, (33, 2, 3))) and not out[1, 33:].any() and not out[1, :, 2:].any()


def test_product_host_transforms_equal_oracle_front_end():
    """the product's host-side transforms (the fallback of the device front end; what bench.py feeds `value_incl_h2d_f32`) are the same
    functions as the oracle's front end, bit for bit"""
    from isegmi.maskrcnn import prepare_images
    from isegmi.transforms import yolact_transform
    rng = np.random.default_rng(11)
    for hw in ((200, 200), (123, 171), (480, 640), (37, 29)):
        x = rng.integers(0, 256, (2,) + hw + (3,), dtype=np.uint8)
        assert np.array_equal(ora.fast_base_transform(x, 200), np.concatenate([yolact_transform(im, 200) for im in x]))
        assert np.array_equal(ora.fast_base_transform(x, 200, darknet=True), np.concatenate([yolact_transform(im, 200, darknet=True) for im in x]))
    ims = [rng.integers(0, 256, (200, 333, 3), dtype=np.uint8), rng.integers(0, 256, (256, 190, 3), dtype=np.uint8)]
    a, hw = ora.to_image_list(ims)
    b, hw2 = prepare_images([im.astype(np.float32) for im in ims])
    assert np.array_equal(a, b) and np.array_equal(hw, hw2)


@pytest.mark.parametrize("name", ["conv", "conv1x1s2", "detmath", "nms", "roi_align", "yolact", "paste", "deform", "frontend"])
def test_oracle_reproduces_golden(name):
    g = gold(name)
    if name == "frontend":
        assert np.array_equal(ora.fast_base_transform(g["y_in"], 64), g["y_out64"]) and np.array_equal(ora.fast_base_transform(g["y_in"], 24), g["y_out24"])
        assert np.array_equal(ora.fast_base_transform(g["y_in"], 64, darknet=True), g["y_dark"])
        out, hw = ora.to_image_list([g["m_in0"], g["m_in1"]])
        assert np.array_equal(out, g["m_out"]) and np.array_equal(hw, g["m_hw"])
    elif name == "conv":
        assert np.array_equal(ora.conv2d(g["x"], g["w"], 1, 1, g["scale"], g["shift"], g["res"], 1), g["y"])
    elif name == "conv1x1s2":
        assert np.array_equal(ora.conv2d(g["x"], g["w"], 2, 0), g["y"])
    elif name == "detmath":
        for i, k in enumerate(("exp", "sigmoid", "tanh")):
            assert np.array_equal(ora.map_f32(g["x"], i), g[k])
        assert np.array_equal(ora.map_f32(np.abs(g["x"]) + np.float32(1e-3), 3), g["log2"])
    elif name == "nms":
        assert np.array_equal(ora.nms(g["boxes"], g["scores"], 0.5, 1, 0), g["keep_gt"])
        assert np.array_equal(ora.nms(g["boxes"], g["scores"], 0.5, 1, 1), g["keep_ge"])
        assert np.array_equal(ora.nms(g["boxes"], g["scores"], 0.5, 0, 0), g["keep_noplus"])
    elif name == "roi_align":
        assert np.array_equal(ora.roi_align(g["feat"], g["rois"], 0.125, 7, 7, 2), g["out7"])
        assert np.array_equal(ora.level_map(g["rois"][:, 1:]), g["levels"])
    elif name == "yolact":
        bx = ora.yolact_decode(g["loc"], g["priors"])
        assert np.array_equal(bx, g["boxes"])
        d = ora.yolact_detect(ora.softmax(g["conf"]), bx, g["mask"])
        assert np.array_equal(d["prior"], g["det_prior"]) and np.array_equal(d["score"], g["det_score"]) and np.array_equal(d["cls"], g["det_cls"])
        mm, ib = ora.yolact_masks(g["proto"], d["mask"], d["box"], 50, 60)
        assert np.array_equal(np.packbits(mm), g["masks"]) and np.array_equal(ib, g["int_boxes"])
    elif name == "paste":
        assert np.array_equal(np.packbits(ora.paste_masks(g["masks"], g["boxes"], 120, 160)), g["out"])
    elif name == "deform":
        assert np.array_equal(ora.deform_im2col(g["x"], g["om"], 3, 3, 2, 1, 1), g["col"])
