"""Randomised HIP-vs-oracle parity sweep over the kernels with the most intricate control flow (dev tool, GPU):
16x16x4 / 32x64 conv tiles on random shapes, chip-wide and single-block RPN NMS, per-class box NMS with crowded classes,
top-k at random sizes, Yolact Detect (softmax / decode / fast-NMS / final top-k) and mask assembly, DCNv2 sampling.
Prints one line per family; exits non-zero on the first mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi as ffi
from isegmi.maskrcnn import generate_anchors, grid_anchors
from oracle import ora

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)

def fail(msg):
    print("MISMATCH", msg); sys.exit(1)

# ---- conv, forced tiles
for it in range(rounds):
    N = int(rng.integers(1, 3)); H = int(rng.integers(3, 40)); W = int(rng.integers(3, 40))
    # round 3: up to 640 input channels (the K walk's 128-channel groups, ragged last group) and up to 1100 outputs (banded tile walk, narrow last band)
    Cin = 32 * int(rng.integers(1, 5) if it % 2 == 0 else rng.integers(5, 21)); Cout = int(rng.choice([5, 16, 31, 32, 33, 64, 65, 96, 130, 200, 260, 352, 416, 700, 1100]))
    R = int(rng.choice([1, 3])); stride = int(rng.choice([1, 2])); pad = R // 2 if rng.uniform() < 0.8 else 0
    if H + 2 * pad < R or W + 2 * pad < R: continue
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32); w = (rng.standard_normal((Cout, R, R, Cin)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32); sh = rng.standard_normal(Cout).astype(np.float32)
    Ho = (H + 2 * pad - R) // stride + 1; Wo = (W + 2 * pad - R) // stride + 1
    res = rng.standard_normal((N, Ho, Wo, Cout)).astype(np.float32) if rng.uniform() < 0.5 else None
    act = int(rng.integers(0, 5))
    if act == 4 and res is None: act = 3
    ref = ora.conv2d(x, w, stride, pad, sc, sh, res, act)
    for tile in (0, 3, 4, 5, 6, 10, 12, 13, 14):
        got = ffi.conv2d(x, w, stride, pad, sc, sh, res, act, tile)
        if not np.array_equal(got, ref): fail(("conv", it, tile, x.shape, w.shape, stride, pad, act))
print("conv ok", rounds)

# ---- conv, grids large enough (> 512 64x64 tiles with a ragged remainder) for the hybrid launch (tiles 13 / 14, and the auto rule) to split
for it in range(max(rounds // 8, 3)):
    N = int(rng.integers(1, 3)); H = int(rng.integers(120, 200)); W = int(rng.integers(120, 200))
    Cin = 32 * int(rng.integers(1, 3)); Cout = int(rng.choice([64, 65, 96, 128])); R = int(rng.choice([1, 3])); stride = 1; pad = R // 2
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32); w = (rng.standard_normal((Cout, R, R, Cin)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, Cout).astype(np.float32); sh = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((N, H, W, Cout)).astype(np.float32) if rng.uniform() < 0.5 else None
    act = int(rng.integers(0, 2))
    ref = ora.conv2d(x, w, stride, pad, sc, sh, res, act)
    for tile in (0, 13, 14):
        got = ffi.conv2d(x, w, stride, pad, sc, sh, res, act, tile)
        if not np.array_equal(got, ref): fail(("conv-hybrid", it, tile, x.shape, w.shape, act))
print("conv hybrid ok")

# ---- RPN level: chip-wide vs single-block vs oracle
for it in range(rounds):
    N = int(rng.integers(1, 3)); H = int(rng.integers(6, 44)); W = int(rng.integers(6, 60)); A = 3
    head = np.concatenate([rng.normal(-2, 2, (N, H, W, A)), rng.normal(0, rng.choice([0.02, 0.3, 1.0]), (N, H, W, 4 * A))], -1).astype(np.float32)
    if rng.uniform() < 0.5: head[0, : H // 2, : W // 2, :A] = 1.0
    anchors = grid_anchors(H, W, 8, generate_anchors(8, int(rng.choice([32, 64, 128])), (0.5, 1.0, 2.0)))
    hw = np.array([[H * 8 - int(rng.integers(0, 8)), W * 8 - int(rng.integers(0, 8))] for _ in range(N)], np.int32)
    pre = int(rng.choice([17, 64, 65, 128, 300, 640, 1000, 1024])); post = int(rng.integers(1, pre + 1))
    ms = float(rng.choice([0.0, 0.0, 16.0, 40.0]))
    for cw in (True, False):
        got = ffi.rpn_level(head, anchors, hw, A, pre, post, min_size=ms, chip_wide=cw)
        for n in range(N):
            rb, rs = ora.rpn_level(head[n, ..., :A].reshape(-1), head[n, ..., A:].reshape(-1, 4), anchors, pre, post, 0.7, ms, float(hw[n, 1]), float(hw[n, 0]))
            if not (np.array_equal(got[n][1], rs) and np.array_equal(got[n][0], rb)): fail(("rpn", it, cw, H, W, pre, post, ms, n))
print("rpn ok", rounds)

# ---- box post-processing with crowded classes
def boxes(n, Wd=1333, Hd=800):
    c = rng.uniform(0, 1, (n, 2)) * [Wd, Hd]
    c[n // 2:] = c[: n - n // 2] + rng.normal(0, rng.choice([1.0, 6.0, 30.0]), (n - n // 2, 2))
    wh = np.exp(rng.uniform(np.log(16), np.log(512), (n, 2)))
    return np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, [Wd - 1, Hd - 1, Wd - 1, Hd - 1]).astype(np.float32)
for it in range(max(rounds // 3, 4)):
    N, R, ncls = 2, int(rng.choice([200, 700, 1000])), 81
    logits = rng.normal(0, 1.0, (N, R, ncls)).astype(np.float32); logits[..., 0] += float(rng.choice([1.0, 2.0, 4.0]))
    hot = rng.choice(np.arange(1, ncls), 3, replace=False); logits[..., hot] += float(rng.choice([1.5, 2.5, 4.0]))
    regr = (rng.normal(0, float(rng.choice([0.02, 0.5])), (N, R, 4 * ncls))).astype(np.float32)
    props = np.stack([boxes(R) for _ in range(N)])
    cnt = np.array([R, int(rng.integers(1, R + 1))], np.int32)
    hw = np.array([[800, 1333], [750, 1200]], np.int32)
    got = ffi.box_postprocess(logits, regr, props, cnt, hw)
    for n in range(N):
        k = cnt[n]
        rb, rs, rl = ora.box_postprocess(logits[n, :k], regr[n, :k], props[n, :k], float(hw[n, 1]), float(hw[n, 0]), cap=100)
        if not (np.array_equal(got[n][2], rl) and np.array_equal(got[n][1], rs) and np.array_equal(got[n][0], rb)): fail(("boxpost", it, R, n))
print("box_postprocess ok", max(rounds // 3, 4))

# ---- top-k
for it in range(rounds):
    rows = int(rng.choice([1, 2, 5, 80, 300])); n = int(rng.choice([50, 777, 4096, 9000, 19248, 70000])); k = int(rng.choice([1, 100, 128, 200, 256, 1000, 1024]))
    if rows * n > 6_000_000: rows = 2
    keys = rng.uniform(0, 1, (rows, n)).astype(np.float32)
    if rng.uniform() < 0.5: keys = np.round(keys * 50) / 50  # heavy ties
    vals, idx, cnt = ffi.topk(keys, k)
    for r in range(min(rows, 6)):
        s, i = ora.topk(keys[r], k)
        if not (cnt[r] == len(s) and np.array_equal(idx[r, : cnt[r]], i) and np.array_equal(vals[r, : cnt[r]], s)): fail(("topk", it, rows, n, k, r))
print("topk ok", rounds)

# ---- Yolact Detect + masks
for it in range(max(rounds // 3, 4)):
    N = int(rng.integers(1, 4)); P = int(rng.choice([300, 1200, 4800, 19248])); ncls = 81
    conf = rng.standard_normal((N, P, ncls)).astype(np.float32); conf[..., 0] += float(rng.choice([2.0, 4.0, 6.0]))
    hotm = rng.uniform(0, 1, (N, P)) < float(rng.choice([0.0, 0.01, 0.05, 0.3]))
    cls = rng.integers(1, ncls, (N, P)) if rng.uniform() < 0.7 else np.full((N, P), 7)
    for n in range(N):
        idx = np.nonzero(hotm[n])[0]
        conf[n, idx, cls[n, idx]] += float(rng.choice([5.0, 8.0]))
    c = rng.uniform(0.05, 0.95, (P, 2)); wh = rng.uniform(0.02, 0.5, (P, 2))
    if rng.uniform() < 0.5: c[P // 2:] = c[: P - P // 2] + rng.normal(0, 0.01, (P - P // 2, 2))
    priors = np.concatenate([c, wh], 1).astype(np.float32)
    loc = (rng.standard_normal((N, P, 4)) * float(rng.choice([0.1, 0.5]))).astype(np.float32)
    mask = np.tanh(rng.standard_normal((N, P, 32))).astype(np.float32)
    got, bx = ffi.yolact_detect(conf, loc, mask, priors)
    for n in range(N):
        rbx = ora.yolact_decode(loc[n], priors)
        ref = ora.yolact_detect(ora.softmax(conf[n]), rbx, mask[n])
        if not np.array_equal(bx[n], rbx): fail(("yolact boxes", it, n))
        for key in ("prior", "cls", "score", "box", "mask"):
            if not np.array_equal(got[n][key], ref[key]): fail(("yolact detect", it, N, P, n, key))
print("yolact detect ok", max(rounds // 3, 4))
for it in range(max(rounds // 6, 2)):
    N, K = int(rng.integers(1, 3)), 100; PH = int(rng.choice([24, 69, 138])); h = int(rng.integers(20, 300)); w = int(rng.integers(20, 300))
    proto = np.maximum(rng.standard_normal((N, PH, PH, 32)), 0).astype(np.float32)
    coeffs = np.tanh(rng.standard_normal((N, K, 32))).astype(np.float32)
    cc = rng.uniform(-0.1, 1.1, (N, K, 2)); ss = rng.uniform(0.01, 0.8, (N, K, 2))
    boxes_ = np.concatenate([cc - ss / 2, cc + ss / 2], -1).astype(np.float32)
    counts = rng.integers(0, K + 1, N).astype(np.int32)
    masks, ib = ffi.yolact_masks(proto, coeffs, boxes_, counts, h, w)
    for n in range(N):
        k = int(counts[n])
        if k == 0: continue
        rm, rb = ora.yolact_masks(proto[n], coeffs[n, :k], boxes_[n, :k], h, w)
        if not (np.array_equal(masks[n, :k], rm) and np.array_equal(ib[n, :k], rb)): fail(("yolact masks", it, n, PH, h, w))
print("yolact masks ok", max(rounds // 6, 2))

# ---- DCNv2 sampling
for it in range(max(rounds // 3, 4)):
    N = int(rng.integers(1, 3)); H = int(rng.integers(3, 30)); W = int(rng.integers(3, 30)); C = 4 * int(rng.integers(1, 20))
    stride = int(rng.choice([1, 2])); dil = int(rng.choice([1, 1, 2])); pad = dil
    Ho = (H + 2 * pad - dil * 2 - 1) // stride + 1; Wo = (W + 2 * pad - dil * 2 - 1) // stride + 1
    if Ho < 1 or Wo < 1: continue
    x = rng.standard_normal((N, H, W, C)).astype(np.float32)
    om = rng.normal(0, float(rng.choice([0.3, 2.0, 8.0])), (N, Ho, Wo, 27)).astype(np.float32)
    if rng.uniform() < 0.5: om[..., :18] = np.round(om[..., :18] * 2) / 2
    if not np.array_equal(ffi.deform_im2col(x, om, 3, 3, stride, pad, dil), ora.deform_im2col(x, om, 3, 3, stride, pad, dil)): fail(("deform", it, x.shape, stride, dil))
print("deform_im2col ok", max(rounds // 3, 4))

# ---- round 5: grouped conv launches (random member count, shapes, kinds of destination), bit-exact per member
for it in range(max(rounds // 3, 4)):
    n = int(rng.integers(1, 8))
    items, refs = [], []
    big = rng.uniform() < 0.4   # now and then a member large enough for the v2-tile kind inside the same launch
    for m in range(n):
        N = int(rng.integers(1, 4)); H = int(rng.integers(3, 30)); W = int(rng.integers(3, 30))
        if big and m == 0: N, H, W = 4, int(rng.integers(90, 130)), int(rng.integers(90, 130))
        Cin = 32 * int(rng.integers(1, 9)); Cout = int(rng.choice([5, 16, 32, 33, 64, 96, 130, 256, 351]))
        R = int(rng.choice([1, 3])); stride = int(rng.choice([1, 1, 2])); pad = R // 2
        x = rng.standard_normal((N, H, W, Cin)).astype(np.float32); w = (rng.standard_normal((Cout, R, R, Cin)) * 0.1).astype(np.float32)
        Ho = (H + 2 * pad - R) // stride + 1; Wo = (W + 2 * pad - R) // stride + 1
        itm = dict(x=x, w=w, stride=stride, pad=pad, act=int(rng.integers(0, 2)))
        if rng.uniform() < 0.7: itm["scale"] = rng.uniform(0.5, 1.5, Cout).astype(np.float32); itm["shift"] = rng.standard_normal(Cout).astype(np.float32)
        if rng.uniform() < 0.4: itm["residual"] = rng.standard_normal((N, Ho, Wo, Cout)).astype(np.float32)
        items.append(itm)
        refs.append(ora.conv2d(x, w, stride, pad, itm.get("scale"), itm.get("shift"), itm.get("residual"), itm["act"]))
    got = ffi.conv2d_group(items)
    for m in range(n):
        if not np.array_equal(got[m], refs[m]): fail(("conv group", it, m, items[m]["x"].shape, items[m]["w"].shape))
print("conv group ok", max(rounds // 3, 4))

# ---- round 5: RPN selection batched over (level, image) against the oracle, level by level
for it in range(max(rounds // 6, 3)):
    N = int(rng.integers(1, 4)); A = 3; nl = int(rng.integers(1, 6))
    heads, ancs = [], []
    for l in range(nl):
        H = int(rng.integers(3, 110 >> l) + 3); W = int(rng.integers(3, 130 >> l) + 3); stride = 4 << l
        h = np.concatenate([rng.normal(-2, 2, (N, H, W, A)), rng.normal(0, 0.3, (N, H, W, 4 * A))], -1).astype(np.float32)
        if rng.uniform() < 0.5: h[0, :2, :3, :A] = 0.75   # ties
        heads.append(h); ancs.append(grid_anchors(H, W, stride, generate_anchors(stride, 8 * stride, (0.5, 1.0, 2.0))))
    hw = np.array([[int(rng.integers(200, 500)), int(rng.integers(200, 600))] for _ in range(N)], np.int32)
    pre = int(rng.integers(257, 1025)); post = int(rng.integers(1, pre + 1)); ms = float(rng.choice([0.0, 0.0, 16.0]))
    got = ffi.rpn_levels(heads, ancs, hw, A, pre, post, min_size=ms)
    for l in range(nl):
        for n_ in range(N):
            rb, rs = ora.rpn_level(heads[l][n_, ..., :A].reshape(-1), heads[l][n_, ..., A:].reshape(-1, 4), ancs[l], pre, post, 0.7, ms, float(hw[n_, 1]), float(hw[n_, 0]))
            if not (np.array_equal(got[l][n_][1], rs) and np.array_equal(got[l][n_][0], rb)): fail(("rpn levels", it, l, n_, pre, post, ms))
print("rpn levels ok", max(rounds // 6, 3))

# ---- round 5: every fp16 tile form of a layer gives the same bits (16x16x32 / 144-row forms against their 32x32x16 twins)
for it in range(max(rounds // 6, 3)):
    N = int(rng.integers(1, 4)); H = int(rng.integers(12, 60)); W = int(rng.integers(12, 90)); Cin = 64 * int(rng.integers(1, 9)); Cout = int(rng.choice([64, 128, 256, 512]))
    R = int(rng.choice([1, 3])); pad = R // 2
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16); w = (rng.standard_normal((Cout, R, R, Cin)) * 0.05).astype(np.float16).astype(np.float32)
    res = rng.standard_normal((N, H, W, Cout)).astype(np.float16) if rng.uniform() < 0.5 else None
    tiles = (30, 40, 41) if R == 3 else (37, 47, 46, 34, 44, 39, 49)
    outs = [ffi.conv2d_f16(x, w, 1, pad, None, None, res, 1, t) for t in tiles]
    for t, o in zip(tiles[1:], outs[1:]):
        if not np.array_equal(o, outs[0]): fail(("f16 tile forms", it, tiles[0], t, x.shape, w.shape))
print("f16 tile forms ok", max(rounds // 6, 3))

# ---- round 5: RoIAlign from roi_prep's table (channel-slice launch, any order) against the plain launch and, on a sample of RoIs, the oracle
for it in range(max(rounds // 6, 3)):
    N = int(rng.integers(1, 4)); K = int(rng.choice([1, 7, 100, 333, 1000])); Cc = int(rng.choice([64, 128, 256])); PH = int(rng.choice([7, 14])); f16 = bool(rng.integers(0, 2))
    Hc = int(rng.integers(40, 210)); Wc = int(rng.integers(40, 340))
    shapes = [(max(Hc >> l, 1), max(Wc >> l, 1)) for l in range(4)]; scales = [0.25, 0.125, 0.0625, 0.03125]
    dt = np.float16 if f16 else np.float32
    feats = [rng.standard_normal((N, h, w, Cc)).astype(dt) for h, w in shapes]
    rois = np.stack([boxes(K, Wc * 4 + 40, Hc * 4 + 40) - 20 for _ in range(N)]).astype(np.float32)   # some hang over every edge
    cnt = rng.integers(0, K + 1, N).astype(np.int32); cnt[0] = K
    plain = ffi.roi_align_f16(feats, scales, rois, cnt, PH, PH) if f16 else ffi.roi_align(feats, scales, rois, cnt, PH, PH)[0]
    order, tab = ffi.roi_prep(rois, cnt, shapes, scales, Cc, PH, PH, f16=f16)
    if sorted(order.reshape(-1).tolist()) != list(range(N * K)): fail(("roi_prep order", it, N, K))
    for o in (order, None, rng.permutation(N * K).astype(np.int32).reshape(N, K)):
        got = ffi.roi_align_ordered(feats, scales, rois, cnt, PH, PH, o, tab, f16=f16)
        if not np.array_equal(got, plain): fail(("roi_align table", it, N, K, Cc, PH, f16))
    lv = ora.level_map(rois[0])
    for k in rng.choice(K, min(K, 5), replace=False):
        L = int(lv[k]); r5 = np.concatenate([[0.0], rois[0, k]]).astype(np.float32)[None]
        ref = ora.roi_align(feats[L - 2].astype(np.float32), r5, scales[L - 2], PH, PH, 2).astype(dt)
        if not np.array_equal(plain.reshape(N, K, PH, PH, Cc)[0, k], ref[0]): fail(("roi_align oracle", it, k))
print("roi_align table ok", max(rounds // 6, 3))
