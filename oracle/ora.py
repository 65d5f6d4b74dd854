"""ctypes binding of the CPU oracle (oracle/ora_ops.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the product package.

PARITY UNPINNED: the reference tree holds no source, tests or golden vectors
for this path (SURVEY.md sections 0 and 8c); the oracle restates SURVEY.md
Appendix A.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "ora_ops.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"] if force else ["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.ora_expf.restype = C.c_float
        _lib.ora_expf.argtypes = [C.c_float]
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


F = C.c_float
I = C.c_int
L = C.c_int64


def map_f32(x, fn):
    """fn: 0 exp, 1 sigmoid, 2 tanh, 3 log2"""
    x = _f(x)
    y = np.empty_like(x)
    lib().ora_map_f32(_p(x), _p(y), L(x.size), I(fn))
    return y


def conv_split_qualifies(M, Cout, R, S, Cin):
    """The shape rule of the split-K mode (the product's isegmi_conv_split_qualifies): at most 176 tiles of 64 x 64 and at least 32 K chunks of 32."""
    return Cin % 32 == 0 and -(-M // 64) * -(-Cout // 64) <= 176 and R * S * (Cin // 32) >= 32


def conv2d(x, w, stride=1, pad=0, scale=None, shift=None, residual=None, act=0, out=None,
           out_img_stride=None, out_pix_stride=None, ksplit=1):
    """x [N,H,W,Cin] NHWC, w [Cout,R,S,Cin]; returns [N,Ho,Wo,Cout] (or writes into `out`).  ksplit > 1: the fixed-tree split-K evaluation
    (ora_conv2d_split: ((p0 + p1) + ...) of ksplit k-ordered partial chains over equal chunk ranges)."""
    x = _f(x); w = _f(w)
    N, H, W_, Cin = x.shape
    Cout, R, S, Cin2 = w.shape
    assert Cin == Cin2
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W_ + 2 * pad - S) // stride + 1
    scale = None if scale is None else _f(scale)
    shift = None if shift is None else _f(shift)
    residual = None if residual is None else _f(residual)
    ret = None
    if out is None:
        ret = out = np.empty((N, Ho, Wo, Cout), np.float32)
        out_img_stride = Ho * Wo * Cout
        out_pix_stride = Cout
    lib().ora_conv2d_split(_p(x), I(N), I(H), I(W_), I(Cin), _p(w), I(Cout), I(R), I(S), I(stride), I(pad),
                           _p(scale), _p(shift), _p(residual), I(act), _p(out), L(out_img_stride), L(out_pix_stride), I(ksplit))
    return ret


def deconv2x2(x, w, bias, relu):
    x = _f(x); w = _f(w); bias = _f(bias)
    N, H, W_, Cin = x.shape
    Cout = w.shape[1]
    out = np.empty((N, 2 * H, 2 * W_, Cout), np.float32)
    lib().ora_deconv2x2(_p(x), I(N), I(H), I(W_), I(Cin), _p(w), I(Cout), _p(bias), I(relu), _p(out))
    return out


def maxpool(x, k, s, p):
    x = _f(x)
    N, H, W_, Cc = x.shape
    Ho = (H + 2 * p - k) // s + 1
    Wo = (W_ + 2 * p - k) // s + 1
    out = np.empty((N, Ho, Wo, Cc), np.float32)
    lib().ora_maxpool(_p(x), I(N), I(H), I(W_), I(Cc), I(k), I(s), I(p), _p(out))
    return out


def resize_bilinear(x, Ho, Wo, add=None, relu=0):
    x = _f(x)
    N, H, W_, Cc = x.shape
    add = None if add is None else _f(add)
    out = np.empty((N, Ho, Wo, Cc), np.float32)
    lib().ora_resize_bilinear(_p(x), I(N), I(H), I(W_), I(Cc), I(Ho), I(Wo), _p(add), I(relu), _p(out))
    return out


def deform_im2col(x, om, R=3, S=3, stride=1, pad=1, dil=1):
    """x [N,H,W,C], om [N,Ho,Wo,3*R*S] (dy/dx interleaved, then mask logits) -> columns [N,Ho,Wo,R*S*C]."""
    x = _f(x); om = _f(om)
    N, H, W_, Cc = x.shape
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1; Wo = (W_ + 2 * pad - dil * (S - 1) - 1) // stride + 1
    assert om.shape == (N, Ho, Wo, 3 * R * S), (om.shape, (N, Ho, Wo, 3 * R * S))
    out = np.empty((N, Ho, Wo, R * S * Cc), np.float32)
    lib().ora_deform_im2col(_p(x), I(N), I(H), I(W_), I(Cc), _p(om), I(R), I(S), I(stride), I(pad), I(dil), _p(out))
    return out


def upsample_nearest2x_add(coarse, lateral):
    coarse = _f(coarse); lateral = _f(lateral)
    N, Hc, Wc, Cc = coarse.shape
    _, H, W_, _ = lateral.shape
    out = np.empty_like(lateral)
    lib().ora_upsample_nearest2x_add(_p(coarse), I(N), I(Hc), I(Wc), I(Cc), _p(lateral), I(H), I(W_), _p(out))
    return out


def softmax(x):
    x = _f(x)
    Cc = x.shape[-1]
    y = np.empty_like(x)
    lib().ora_softmax(_p(x), L(x.size // Cc), I(Cc), _p(y))
    return y


def topk(scores, k):
    scores = _f(scores)
    n = scores.size
    m = min(k, n)
    s = np.empty(max(m, 1), np.float32); i = np.empty(max(m, 1), np.int32)
    c = lib().ora_topk(_p(scores), I(n), I(k), _p(s), _p(i))
    return s[:c], i[:c]


def decode_boxes(anchors, deltas, weights, im_w, im_h, clip=True):
    anchors = _f(anchors); deltas = _f(deltas)
    n = anchors.shape[0]
    out = np.empty((n, 4), np.float32)
    lib().ora_decode_boxes(_p(anchors), _p(deltas), I(n), F(weights[0]), F(weights[1]), F(weights[2]),
                           F(weights[3]), F(im_w), F(im_h), I(1 if clip else 0), _p(out))
    return out


def nms(boxes, scores, thr, plus_one=1, ge=0, max_keep=0):
    boxes = _f(boxes); scores = _f(scores)
    n = scores.size
    keep = np.empty(max(n, 1), np.int32)
    c = lib().ora_nms(_p(boxes), _p(scores), I(n), F(thr), I(plus_one), I(ge), I(max_keep), _p(keep))
    return keep[:c].copy()


def rpn_level(logits, deltas, anchors, pre_nms, post_nms, nms_thr, min_size, im_w, im_h, nms_flags=0):
    """nms_flags: App. A.6 forks, 1 suppress on >=, 2 plain areas (no +1)"""
    logits = _f(logits).ravel(); deltas = _f(deltas).reshape(-1, 4); anchors = _f(anchors)
    hwa = logits.size
    ob = np.empty((max(post_nms, 1), 4), np.float32); os_ = np.empty(max(post_nms, 1), np.float32)
    c = lib().ora_rpn_level(_p(logits), _p(deltas), _p(anchors), I(hwa), I(pre_nms), I(post_nms), F(nms_thr),
                            F(min_size), F(im_w), F(im_h), I(nms_flags), _p(ob), _p(os_))
    return ob[:c].copy(), os_[:c].copy()


def level_map(boxes, k_min=2, k_max=5):
    boxes = _f(boxes)
    n = boxes.shape[0]
    lvl = np.empty(max(n, 1), np.int32)
    lib().ora_level_map(_p(boxes), I(n), I(k_min), I(k_max), _p(lvl))
    return lvl[:n]


def roi_align(feat, rois, spatial_scale, PH, PW, g=2, aligned=0):
    """aligned: App. A.7 fork, 0 the legacy op, 1 ROIAlign(aligned=True)"""
    feat = _f(feat); rois = _f(rois).reshape(-1, 5)
    N, H, W_, Cc = feat.shape
    R = rois.shape[0]
    out = np.empty((R, PH, PW, Cc), np.float32)
    lib().ora_roi_align2(_p(feat), I(N), I(H), I(W_), I(Cc), _p(rois), I(R), F(spatial_scale), I(PH), I(PW), I(g), I(aligned), _p(out))
    return out


def avgpool_full(x):
    """[R,H,W,C] -> [R,C]: AvgPool2d over the whole H x W window (C4 FastRCNNPredictor)."""
    x = _f(x)
    R_, H, W_, Cc = x.shape
    out = np.empty((R_, Cc), np.float32)
    lib().ora_avgpool_full(_p(x), I(R_), I(H * W_), I(Cc), _p(out))
    return out


def box_postprocess(logits, regr, props, im_w, im_h, score_thr=0.05, nms_thr=0.5, det_per_img=100,
                    nms_flags=0, cap=128):
    """nms_flags: App. A.6 forks, 1 suppress on >=, 2 plain areas (no +1), 4 a class's detections in proposal-index order"""
    logits = _f(logits); regr = _f(regr); props = _f(props)
    R, ncls = logits.shape
    ob = np.empty((cap, 4), np.float32); os_ = np.empty(cap, np.float32); ol = np.empty(cap, np.int32)
    c = lib().ora_box_postprocess(_p(logits), _p(regr), _p(props), I(R), I(ncls), F(im_w), F(im_h), F(score_thr),
                                  F(nms_thr), I(det_per_img), I(nms_flags), I(cap), _p(ob), _p(os_), _p(ol))
    return ob[:c].copy(), os_[:c].copy(), ol[:c].copy()


def mask_logits_select(feat, w, b, labels):
    feat = _f(feat); w = _f(w); b = _f(b)
    labels = np.ascontiguousarray(labels, np.int32)
    R = feat.shape[0]; Cc = feat.shape[-1]; HW = feat.size // (R * Cc) if R else 0
    out = np.empty((R, HW), np.float32)
    lib().ora_mask_logits_select(_p(feat), I(R), I(HW), I(Cc), _p(w), _p(b), _p(labels), _p(out))
    return out


def paste_masks(masks, boxes, im_h, im_w, thr=0.5):
    masks = _f(masks); boxes = _f(boxes)
    n = masks.shape[0]; M = masks.shape[-1]
    out = np.empty((n, im_h, im_w), np.uint8)
    lib().ora_paste_masks(_p(masks), _p(boxes), I(n), I(M), I(im_h), I(im_w), F(thr), _p(out))
    return out


def yolact_decode(loc, priors):
    loc = _f(loc); priors = _f(priors)
    P = priors.shape[0]
    out = np.empty((P, 4), np.float32)
    lib().ora_yolact_decode(_p(loc), _p(priors), I(P), _p(out))
    return out


def yolact_detect(conf, boxes, mask, conf_thresh=0.05, nms_thr=0.5, top_k=200, max_det=100, second_threshold=0):
    conf = _f(conf); boxes = _f(boxes); mask = _f(mask)
    P, ncls = conf.shape
    md = mask.shape[1]
    ob = np.empty((max_det, 4), np.float32); os_ = np.empty(max_det, np.float32)
    oc = np.empty(max_det, np.int32); om = np.empty((max_det, md), np.float32); op = np.empty(max_det, np.int32)
    c = lib().ora_yolact_detect2(_p(conf), _p(boxes), _p(mask), I(P), I(ncls), I(md), F(conf_thresh), F(nms_thr),
                                 I(top_k), I(max_det), I(second_threshold), _p(ob), _p(os_), _p(oc), _p(om), _p(op))
    return dict(box=ob[:c].copy(), score=os_[:c].copy(), cls=oc[:c].copy(), mask=om[:c].copy(), prior=op[:c].copy())


def yolact_proto_masks(proto, coeffs, boxes):
    """crop(sigmoid(proto @ coeffs)) at prototype resolution: [n, PH, PW] (the mask-IoU net's input)."""
    proto = _f(proto); coeffs = _f(coeffs); boxes = _f(boxes)
    PH, PW, K = proto.shape
    n = coeffs.shape[0]
    lo = np.empty((max(n, 1), PH, PW), np.float32)
    lib().ora_yolact_proto_masks(_p(proto), I(PH), I(PW), I(K), _p(coeffs), _p(boxes), I(n), _p(lo))
    return lo[:n]


def yolact_masks(proto, coeffs, boxes, h, w):
    proto = _f(proto); coeffs = _f(coeffs); boxes = _f(boxes)
    PH, PW, K = proto.shape
    n = coeffs.shape[0]
    out = np.empty((n, h, w), np.uint8)
    ob = np.empty((n, 4), np.int64)
    lib().ora_yolact_masks(_p(proto), I(PH), I(PW), I(K), _p(coeffs), _p(boxes), I(n), I(h), I(w), _p(out), _p(ob))
    return out, ob


def rle_encode(mask):
    """pycocotools rleEncode restated (ora_rle_encode): HxW binary mask -> column-major run lengths, zeros first (np.uint32)."""
    m = np.ascontiguousarray(mask).astype(np.uint8)
    h, w = m.shape
    out = np.empty(h * w + 1, np.uint32)
    lib().ora_rle_encode.restype = L
    k = lib().ora_rle_encode(_p(m), I(h), I(w), I(w), _p(out))
    return out[:k].copy()


def rle_to_string(counts):
    """pycocotools rleToString restated (ora_rle_to_string) -> str."""
    c = np.ascontiguousarray(counts, np.uint32)
    buf = C.create_string_buffer(7 * c.size + 1)
    lib().ora_rle_to_string.restype = L
    n = lib().ora_rle_to_string(_p(c), L(c.size), buf)
    return buf.raw[:n].decode("ascii")


def set_conv_sum_mode(mode):
    """0: the oracle's k-ordered fmaf chain (default); 1: 16-term partial sums added to the accumulator -- a second, equally valid fp32
    association, used ONLY to measure how far two correct evaluations of the fp16-storage network drift apart (ora_set_conv_sum_mode)."""
    lib().ora_set_conv_sum_mode(I(int(mode)))


# ---- front end: uint8 BGR images in, the networks' fp32 NHWC input out
YOLACT_MEANS = (103.94, 116.78, 123.68)   # BGR, SURVEY 8a Y1
YOLACT_STD = (57.38, 57.12, 58.40)
PIXEL_MEAN = (102.9801, 115.9465, 122.7717)  # BGR, SURVEY App. A.0 (INPUT.PIXEL_MEAN; PIXEL_STD (1, 1, 1), TO_BGR255)


def fast_base_transform(images_u8, size=550, darknet=False):
    """FastBaseTransform (Y1): [N, H, W, 3] uint8 BGR -> [N, size, size, 3] fp32 RGB (ora_fast_base_transform)."""
    x = np.ascontiguousarray(images_u8)
    assert x.dtype == np.uint8 and x.ndim == 4 and x.shape[3] == 3
    n, h, w, _ = x.shape
    mean = _f((0.0, 0.0, 0.0) if darknet else YOLACT_MEANS)
    std = _f((255.0, 255.0, 255.0) if darknet else YOLACT_STD)
    out = np.empty((n, size, size, 3), np.float32)
    lib().ora_fast_base_transform(_p(x), I(n), I(h), I(w), I(size), _p(mean), _p(std), I(1), _p(out))
    return out


def to_image_list(images_u8, divisibility=32):
    """build_transform (after the PIL resize) + to_image_list (M1): list of [h, w, 3] uint8 BGR -> ([N, Hpad, Wpad, 3] fp32, image_sizes [N, 2])."""
    hw = np.array([im.shape[:2] for im in images_u8], np.int32)
    H = int(-(-int(hw[:, 0].max()) // divisibility) * divisibility)
    W = int(-(-int(hw[:, 1].max()) // divisibility) * divisibility)
    out = np.empty((len(images_u8), H, W, 3), np.float32)
    mean = _f(PIXEL_MEAN)
    for i, im in enumerate(images_u8):
        a = np.ascontiguousarray(im)
        assert a.dtype == np.uint8 and a.ndim == 3 and a.shape[2] == 3
        lib().ora_build_transform(_p(a), I(a.shape[0]), I(a.shape[1]), I(H), I(W), _p(mean), _p(out[i]))
    return out, hw
