"""ctypes binding of libisegmi.so (C ABI in include/isegmi.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails, an
exception is raised.  Nothing here imports torch, triton or the test oracle.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "libisegmi.so")


class IsegmiError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib, LIB_PATH
    if _lib is None:
        if os.environ.get("ISEGMI_LIB"):  # development hook: same-box A/B of two builds of libisegmi.so (tools/ab_f16.sh)
            LIB_PATH = os.environ["ISEGMI_LIB"]
        if not os.path.exists(LIB_PATH):
            raise IsegmiError(
                "libisegmi.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C instancesegmentation-jittor_amd`; there is no CPU fallback" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.isegmi_last_error.restype = C.c_char_p
        _lib.isegmi_op_roi_table_bytes.restype = C.c_int64
    return _lib


def check(rc):
    if rc != 0:
        raise IsegmiError("libisegmi error %d: %s" % (rc, lib().isegmi_last_error().decode(errors="replace")))


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N", "H", "W", "Cin", "Cout", "R", "S", "stride", "pad", "act", "tile", "out_div")] + \
               [("out_img_stride", C.c_int64), ("out_pix_stride", C.c_int64)]


def device_count():
    n = C.c_int(0)
    rc = lib().isegmi_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def set_device(i):
    check(lib().isegmi_set_device(C.c_int(i)))


def sync():
    check(lib().isegmi_sync())


def set_f16_mfma_shape(shape):
    """0: v_mfma_f32_32x32x16_f16 everywhere; 1: row strips + fused RPN head on v_mfma_f32_16x16x32_f16; 2: persistent tiles too; 3 (default): 1 + the
    144-row tiles (process-wide; isegmi_set_f16_mfma_shape).  Results are bit-identical under every setting."""
    check(lib().isegmi_set_f16_mfma_shape(int(shape)))


def get_f16_mfma_shape():
    v = C.c_int()
    check(lib().isegmi_get_f16_mfma_shape(C.byref(v)))
    return v.value


def box_calibrate(ms_per_leg=30.0):
    """{mfma_f32_tflops, mfma_f16_tflops, hbm_copy_gbs} of bare loops on the current device (isegmi_box_calibrate)."""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    check(lib().isegmi_box_calibrate(C.c_double(ms_per_leg), C.byref(a), C.byref(b), C.byref(c)))
    return {"mfma_f32_tflops": round(a.value, 1), "mfma_f16_tflops": round(b.value, 1), "hbm_copy_gbs": round(c.value, 1)}


class DeviceBuffer:
    """Owning device allocation with numpy shape/dtype metadata."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().isegmi_malloc(C.byref(p), C.c_int64(self.nbytes)))
        self.ptr = p

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.shape, a.dtype)
        if b.nbytes:
            check(lib().isegmi_h2d(b.ptr, a.ctypes.data_as(C.c_void_p), C.c_int64(b.nbytes)))
        return b

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        check(lib().isegmi_h2d(self.ptr, a.ctypes.data_as(C.c_void_p), C.c_int64(self.nbytes)))

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if self.nbytes:
            check(lib().isegmi_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, C.c_int64(self.nbytes)))
        return out

    def zero(self):
        check(lib().isegmi_memset(self.ptr, C.c_int(0), C.c_int64(self.nbytes)))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            lib().isegmi_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedBuffer:
    """Page-locked host memory with a numpy view (`.array`): the source of the engines' asynchronous uploads."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        check(lib().isegmi_malloc_host(C.byref(p), C.c_int64(self.nbytes)))
        self.ptr = p
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(p.value), self.dtype).reshape(self.shape)

    def free(self):
        if self.ptr is not None and self.ptr.value:
            self.array = None
            lib().isegmi_free_host(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _ptr(b):
    return None if b is None else b.ptr


def make_conv_desc(N, H, W, Cin, Cout, R, S, stride=1, pad=0, act=0, tile=0, out_div=0, out_img_stride=0,
                   out_pix_stride=0):
    return ConvDesc(N, H, W, Cin, Cout, R, S, stride, pad, act, tile, out_div, out_img_stride, out_pix_stride)


def conv_out_hw(desc):
    ho, wo = C.c_int32(), C.c_int32()
    check(lib().isegmi_conv_out_hw(C.byref(desc), C.byref(ho), C.byref(wo)))
    return ho.value, wo.value


def pack_conv_weights(desc, w_krsc):
    """host: [Cout,R,S,Cin] fp32 -> packed 1-D fp32 array."""
    w = np.ascontiguousarray(w_krsc, np.float32)
    assert w.shape == (desc.Cout, desc.R, desc.S, desc.Cin), (w.shape, desc.Cout, desc.R, desc.S, desc.Cin)
    n = C.c_int64()
    check(lib().isegmi_conv_packed_floats(C.byref(desc), C.byref(n)))
    out = np.empty(n.value, np.float32)
    check(lib().isegmi_pack_conv_weights(C.byref(desc), w.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    return out


def op_conv2d(desc, d_in, d_wpacked, d_scale=None, d_shift=None, d_residual=None, d_out=None):
    check(lib().isegmi_op_conv2d(C.byref(desc), _ptr(d_in), _ptr(d_wpacked), _ptr(d_scale), _ptr(d_shift),
                                 _ptr(d_residual), _ptr(d_out), None))


def conv2d(x, w_krsc, stride=1, pad=0, scale=None, shift=None, residual=None, act=0, tile=0):
    """Convenience host->device->host convolution (tests / small inputs)."""
    x = np.ascontiguousarray(x, np.float32)
    N, H, W, Cin = x.shape
    Cout, R, S, _ = w_krsc.shape
    d = make_conv_desc(N, H, W, Cin, Cout, R, S, stride, pad, act, tile)
    ho, wo = conv_out_hw(d)
    dx = DeviceBuffer.from_numpy(x)
    dw = DeviceBuffer.from_numpy(pack_conv_weights(d, w_krsc))
    ds = None if scale is None else DeviceBuffer.from_numpy(np.asarray(scale, np.float32))
    dh = None if shift is None else DeviceBuffer.from_numpy(np.asarray(shift, np.float32))
    dr = None if residual is None else DeviceBuffer.from_numpy(np.asarray(residual, np.float32))
    do = DeviceBuffer((N, ho, wo, Cout))
    op_conv2d(d, dx, dw, ds, dh, dr, do)
    return do.numpy()


def conv2d_group(items):
    """isegmi_op_conv2d_group: several independent fp32 convolutions as ONE launch.  items: dicts with x, w (KRSC) and optionally stride, pad, scale,
    shift, residual, act.  -> list of outputs (each what conv2d gives for the member alone, bit for bit)."""
    n = len(items)
    descs = (ConvDesc * n)()
    keep, ins, ws, scs, shs, rss, outs, shapes = [], [], [], [], [], [], [], []
    for i, it in enumerate(items):
        x = np.ascontiguousarray(it["x"], np.float32)
        N, H, W, Cin = x.shape
        Cout, R, S, _ = it["w"].shape
        d = make_conv_desc(N, H, W, Cin, Cout, R, S, it.get("stride", 1), it.get("pad", 0), it.get("act", 0), 0)
        descs[i] = d
        ho, wo = conv_out_hw(d)
        dx = DeviceBuffer.from_numpy(x); dw = DeviceBuffer.from_numpy(pack_conv_weights(d, it["w"]))
        ds = None if it.get("scale") is None else DeviceBuffer.from_numpy(np.asarray(it["scale"], np.float32))
        dh = None if it.get("shift") is None else DeviceBuffer.from_numpy(np.asarray(it["shift"], np.float32))
        dr = None if it.get("residual") is None else DeviceBuffer.from_numpy(np.asarray(it["residual"], np.float32))
        do = DeviceBuffer((N, ho, wo, Cout))
        keep += [dx, dw, ds, dh, dr, do]
        for lst, b in ((ins, dx), (ws, dw), (scs, ds), (shs, dh), (rss, dr), (outs, do)):
            lst.append(None if b is None else b.ptr.value)
    arr = lambda lst: (C.c_void_p * n)(*lst)
    check(lib().isegmi_op_conv2d_group(n, descs, arr(ins), arr(ws), arr(scs), arr(shs), arr(rss), arr(outs), None))
    return [keep[6 * i + 5].numpy() for i in range(n)]


def maxpool(x, k, s, p):
    x = np.ascontiguousarray(x, np.float32)
    N, H, W, Cc = x.shape
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dx = DeviceBuffer.from_numpy(x); do = DeviceBuffer((N, ho, wo, Cc))
    check(lib().isegmi_op_maxpool(dx.ptr, N, H, W, Cc, k, s, p, do.ptr, None))
    return do.numpy()


def resize_bilinear(x, Ho, Wo, add=None, relu=0):
    x = np.ascontiguousarray(x, np.float32)
    N, H, W, Cc = x.shape
    dx = DeviceBuffer.from_numpy(x); do = DeviceBuffer((N, Ho, Wo, Cc))
    da = None if add is None else DeviceBuffer.from_numpy(np.asarray(add, np.float32))
    check(lib().isegmi_op_resize_bilinear(dx.ptr, N, H, W, Cc, Ho, Wo, _ptr(da), relu, do.ptr, None))
    return do.numpy()


def preprocess_u8(images_u8, Hout, Wout, Hpad, Wpad, mean3, std3, swap_rb):
    """isegmi_op_preprocess_u8 (the device front end, M1 / Y1): [N, Hin, Win, 3] uint8 -> [N, Hpad, Wpad, 3] fp32."""
    x = np.ascontiguousarray(images_u8)
    if x.dtype != np.uint8 or x.ndim != 4 or x.shape[3] != 3:
        raise TypeError("preprocess_u8 takes an [N, H, W, 3] uint8 batch")
    N, H, W, _ = x.shape
    dx = DeviceBuffer.from_numpy(x); do = DeviceBuffer((N, Hpad, Wpad, 3))
    m = (C.c_float * 3)(*[float(v) for v in mean3]); sd = (C.c_float * 3)(*[float(v) for v in std3])
    check(lib().isegmi_op_preprocess_u8(dx.ptr, N, H, W, do.ptr, Hout, Wout, Hpad, Wpad, C.c_int64(Hpad * Wpad * 3), m, sd, int(bool(swap_rb)), None))
    return do.numpy()


def deform_im2col(x, om, R=3, S=3, stride=1, pad=1, dil=1):
    """x [N,H,W,C], om [N,Ho,Wo,3*R*S] raw conv_offset_mask output -> DCNv2 columns [N,Ho,Wo,R*S*C]."""
    x = np.ascontiguousarray(x, np.float32); om = np.ascontiguousarray(om, np.float32)
    N, H, W, Cc = x.shape
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1; Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    assert om.shape == (N, Ho, Wo, 3 * R * S), om.shape
    dx = DeviceBuffer.from_numpy(x); dm = DeviceBuffer.from_numpy(om); do = DeviceBuffer((N, Ho, Wo, R * S * Cc))
    check(lib().isegmi_op_deform_im2col(dx.ptr, N, H, W, Cc, dm.ptr, R, S, stride, pad, dil, do.ptr, None))
    return do.numpy()


def upsample_nearest2x_add(coarse, lateral):
    coarse = np.ascontiguousarray(coarse, np.float32); lateral = np.ascontiguousarray(lateral, np.float32)
    N, Hc, Wc, Cc = coarse.shape
    _, H, W, _ = lateral.shape
    dc = DeviceBuffer.from_numpy(coarse); dl = DeviceBuffer.from_numpy(lateral); do = DeviceBuffer(lateral.shape)
    check(lib().isegmi_op_upsample_nearest2x_add(dc.ptr, N, Hc, Wc, Cc, dl.ptr, H, W, do.ptr, None))
    return do.numpy()


def map_f32(x, fn):
    x = np.ascontiguousarray(x, np.float32)
    dx = DeviceBuffer.from_numpy(x); dy = DeviceBuffer(x.shape)
    check(lib().isegmi_op_map_f32(dx.ptr, dy.ptr, C.c_int64(x.size), fn, None))
    return dy.numpy()


def topk(keys2d, k, limit=None, rows_per_limit=1):
    """keys2d [rows][n] -> (vals [rows][k], idx [rows][k], cnt [rows])"""
    keys2d = np.ascontiguousarray(keys2d, np.float32)
    rows, n = keys2d.shape
    dk = DeviceBuffer.from_numpy(keys2d)
    dv = DeviceBuffer((rows, k), np.float32); di = DeviceBuffer((rows, k), np.int32); dc = DeviceBuffer((rows,), np.int32)
    dv.zero(); di.zero()
    dl = None if limit is None else DeviceBuffer.from_numpy(np.asarray(limit, np.int32))
    check(lib().isegmi_op_topk(dk.ptr, C.c_int64(n), rows, n, k, _ptr(dl), rows_per_limit, dv.ptr, di.ptr, dc.ptr, None))
    return dv.numpy(), di.numpy(), dc.numpy()


class YolactDetectArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "P", "ncls", "mask_dim", "top_k", "max_det")] + \
               [("conf_thresh", C.c_float), ("nms_thresh", C.c_float)] + \
               [(n, C.c_void_p) for n in (
                   "d_conf", "d_loc", "d_mask", "d_priors", "d_ws_scoresT", "d_ws_boxes", "d_ws_counts", "d_ws_tk_vals",
                   "d_ws_tk_idx", "d_ws_tk_cnt", "d_ws_cand", "d_ws_fin_vals", "d_ws_fin_idx", "d_ws_fin_cnt",
                   "d_out_count", "d_out_boxes", "d_out_scores", "d_out_classes", "d_out_coeffs", "d_out_prior")] + \
               [("A", C.c_int32), ("mask_tanh", C.c_int32), ("pix_stride", C.c_int64), ("off_loc", C.c_int32), ("off_conf", C.c_int32),
                ("off_mask", C.c_int32), ("second_threshold", C.c_int32)]


def yolact_detect(conf_logits, loc, mask, priors, conf_thresh=0.05, nms_thresh=0.5, top_k=200, max_det=100, second_threshold=0):
    """Host convenience wrapper: conf_logits [N,P,ncls], loc [N,P,4], mask [N,P,md], priors [P,4]."""
    conf_logits = np.ascontiguousarray(conf_logits, np.float32)
    N, P, ncls = conf_logits.shape
    md = mask.shape[-1]
    nc = ncls - 1
    bufs = dict(
        d_conf=DeviceBuffer.from_numpy(conf_logits), d_loc=DeviceBuffer.from_numpy(np.asarray(loc, np.float32)),
        d_mask=DeviceBuffer.from_numpy(np.asarray(mask, np.float32)), d_priors=DeviceBuffer.from_numpy(np.asarray(priors, np.float32)),
        d_ws_scoresT=DeviceBuffer((N, nc, P)), d_ws_boxes=DeviceBuffer((N, P, 4)), d_ws_counts=DeviceBuffer((2 * N,), np.int32),
        d_ws_tk_vals=DeviceBuffer((N, nc, top_k)), d_ws_tk_idx=DeviceBuffer((N, nc, top_k), np.int32),
        d_ws_tk_cnt=DeviceBuffer((N, nc), np.int32), d_ws_cand=DeviceBuffer((N, nc, top_k)),
        d_ws_fin_vals=DeviceBuffer((N, max_det)), d_ws_fin_idx=DeviceBuffer((N, max_det), np.int32),
        d_ws_fin_cnt=DeviceBuffer((N,), np.int32), d_out_count=DeviceBuffer((N,), np.int32),
        d_out_boxes=DeviceBuffer((N, max_det, 4)), d_out_scores=DeviceBuffer((N, max_det)),
        d_out_classes=DeviceBuffer((N, max_det), np.int32), d_out_coeffs=DeviceBuffer((N, max_det, md)),
        d_out_prior=DeviceBuffer((N, max_det), np.int32))
    a = YolactDetectArgs(N, P, ncls, md, top_k, max_det, conf_thresh, nms_thresh, *[bufs[n].ptr for n, _ in YolactDetectArgs._fields_[8:28]])
    a.second_threshold = int(second_threshold)
    check(lib().isegmi_op_yolact_detect(C.byref(a), None))
    cnt = bufs["d_out_count"].numpy()
    out = []
    B, S, Cl, M, Pr = (bufs[k].numpy() for k in ("d_out_boxes", "d_out_scores", "d_out_classes", "d_out_coeffs", "d_out_prior"))
    for n in range(N):
        c = int(cnt[n])
        out.append(dict(box=B[n, :c], score=S[n, :c], cls=Cl[n, :c], mask=M[n, :c], prior=Pr[n, :c]))
    return out, bufs["d_ws_boxes"].numpy()


def yolact_masks(proto, coeffs, boxes, counts, h, w):
    """proto [N,PH,PW,32]; coeffs [N,K,32]; boxes [N,K,4]; counts [N] -> (masks u8 [N,K,h,w], boxes i64 [N,K,4])"""
    proto = np.ascontiguousarray(proto, np.float32)
    N, PH, PW, md = proto.shape
    K = coeffs.shape[1]
    dp = DeviceBuffer.from_numpy(proto); dc = DeviceBuffer.from_numpy(np.asarray(coeffs, np.float32))
    db = DeviceBuffer.from_numpy(np.asarray(boxes, np.float32)); dn = DeviceBuffer.from_numpy(np.asarray(counts, np.int32))
    dlo = DeviceBuffer.from_numpy(np.full((N, K, PH, PW), np.nan, np.float32))   # the workspace arrives dirty: only crop-window pixels may be read back
    dm = DeviceBuffer((N, K, h, w), np.uint8); dm.zero()
    dob = DeviceBuffer((N, K, 4), np.int64)
    check(lib().isegmi_op_yolact_masks(dp.ptr, dc.ptr, db.ptr, dn.ptr, N, PH, PW, md, K, h, w, dlo.ptr, dm.ptr, dob.ptr, None))
    return dm.numpy(), dob.numpy()


# ---------------------------------------------------------------- Mask R-CNN op wrappers (tests / small inputs)
NMS_GE, NMS_NO_PLUS_ONE, NMS_INDEX_ORDER = 1, 2, 4   # ISEGMI_NMS_* of include/isegmi.h (SURVEY App. A.6 forks)


def nms(boxes, scores, thr, plus_one=1, ge=0, max_keep=0):
    """boxes [P,n,4], scores [P,n] -> list of kept original indices (score order) per problem."""
    boxes = np.ascontiguousarray(boxes, np.float32); scores = np.ascontiguousarray(scores, np.float32)
    P, n = scores.shape
    db = DeviceBuffer.from_numpy(boxes); ds = DeviceBuffer.from_numpy(scores)
    dk = DeviceBuffer((P, n), np.int32); dc = DeviceBuffer((P,), np.int32)
    check(lib().isegmi_op_nms(db.ptr, ds.ptr, P, n, C.c_float(thr), plus_one, ge, max_keep, dk.ptr, dc.ptr, None))
    k, c = dk.numpy(), dc.numpy()
    return [k[i, : c[i]].copy() for i in range(P)]


def roi_align(feats, scales, rois, counts, PH, PW, sampling=2, k_min=2, fixed_level=-1, aligned=0):
    """feats: list of [N,H,W,C]; rois [N,K,4]; counts [N] -> (out [N*K,PH,PW,C], levels [N,K])"""
    fb = [DeviceBuffer.from_numpy(np.ascontiguousarray(f, np.float32)) for f in feats]
    N, K = rois.shape[:2]
    Cc = feats[0].shape[3]
    ptrs = (C.c_void_p * len(fb))(*[b.ptr.value for b in fb])
    Hs = (C.c_int32 * len(fb))(*[f.shape[1] for f in feats]); Ws = (C.c_int32 * len(fb))(*[f.shape[2] for f in feats])
    sc = (C.c_float * len(fb))(*scales)
    dr = DeviceBuffer.from_numpy(np.ascontiguousarray(rois, np.float32)); dcnt = DeviceBuffer.from_numpy(np.ascontiguousarray(counts, np.int32))
    do = DeviceBuffer((N * K, PH, PW, Cc)); dl = DeviceBuffer((N, K), np.int32); dl.zero()
    check(lib().isegmi_op_roi_align(ptrs, Hs, Ws, sc, len(fb), dr.ptr, dcnt.ptr, N, K, Cc, PH, PW, sampling, aligned, k_min, fixed_level,
                                    do.ptr, dl.ptr, None))
    return do.numpy(), dl.numpy()


def roi_prep(rois, counts, shapes, scales, Cc, PH, PW, k_min=2, f16=False, want_order=True, aligned=0):
    """first launch of the FPN heads' RoIAlign: rois [N,K,4], counts [N], shapes [(H, W)] per level -> (order [N,K] int32 or None, table [N*K, 2*(PH+PW)+1, 4]
    int32): per-RoI sample rows / columns {low byte offset, high byte offset, low weight bits, high weight bits}, then {level index, H, W, layout signature}"""
    N, K = rois.shape[:2]
    dr = DeviceBuffer.from_numpy(np.ascontiguousarray(rois, np.float32)); dcnt = DeviceBuffer.from_numpy(np.ascontiguousarray(counts, np.int32))
    do = DeviceBuffer((N, K), np.int32) if want_order else None
    TS = 2 * (PH + PW) + 1
    assert lib().isegmi_op_roi_table_bytes(N, K, PH, PW) == N * K * TS * 16
    dtab = DeviceBuffer.from_numpy(np.full((N * K, TS, 4), -1, np.int32))
    Hs = (C.c_int32 * len(shapes))(*[h for h, _ in shapes]); Ws = (C.c_int32 * len(shapes))(*[w for _, w in shapes])
    sc = (C.c_float * len(scales))(*scales)
    check(lib().isegmi_op_roi_prep(dr.ptr, dcnt.ptr, N, K, Hs, Ws, sc, len(scales), k_min, Cc, PH, PW, 2 if f16 else 4, aligned, do.ptr if do else None, dtab.ptr, None))
    return (do.numpy() if do else None), dtab.numpy()


def roi_align_ordered(feats, scales, rois, counts, PH, PW, order, table, k_min=2, f16=False):
    """second launch: RoIAlign (sampling 2, LevelMapper) from roi_prep's `table`, launched in `order` [N,K] (any permutation of the rows, or None = row order)
    -> out [N*K,PH,PW,C], pre-filled with NaN so that a row the launch did not write shows"""
    dt = np.float16 if f16 else np.float32
    fb = [DeviceBuffer.from_numpy(np.ascontiguousarray(f, dt)) for f in feats]
    N, K = rois.shape[:2]
    Cc = feats[0].shape[3]
    ptrs = (C.c_void_p * len(fb))(*[b.ptr.value for b in fb])
    Hs = (C.c_int32 * len(fb))(*[f.shape[1] for f in feats]); Ws = (C.c_int32 * len(fb))(*[f.shape[2] for f in feats])
    sc = (C.c_float * len(fb))(*scales)
    dr = DeviceBuffer.from_numpy(np.ascontiguousarray(rois, np.float32)); dcnt = DeviceBuffer.from_numpy(np.ascontiguousarray(counts, np.int32))
    dord = DeviceBuffer.from_numpy(np.ascontiguousarray(order, np.int32)) if order is not None else None
    dtab = DeviceBuffer.from_numpy(np.ascontiguousarray(table, np.int32))
    do = DeviceBuffer.from_numpy(np.full((N * K, PH, PW, Cc), np.nan, dt))
    fn = lib().isegmi_op_roi_align_f16_ordered if f16 else lib().isegmi_op_roi_align_ordered
    check(fn(ptrs, Hs, Ws, sc, len(fb), dr.ptr, dcnt.ptr, dord.ptr if dord else None, dtab.ptr, N, K, Cc, PH, PW, k_min, do.ptr, None))
    return do.numpy()


def avgpool_full(x):
    """[R,H,W,C] fp32 -> [R,C]: AvgPool2d over the whole window."""
    x = np.ascontiguousarray(x, np.float32)
    R_, H, W, Cc = x.shape
    dx = DeviceBuffer.from_numpy(x); do = DeviceBuffer((R_, Cc))
    check(lib().isegmi_op_avgpool_full(dx.ptr, C.c_int64(R_), H * W, Cc, do.ptr, None))
    return do.numpy()


def roi_align_f16(feats, scales, rois, counts, PH, PW, sampling=2, k_min=2, aligned=0):
    """fp16-storage RoIAlign: feats list of [N,H,W,C] (cast to fp16) -> out [N*K,PH,PW,C] fp16."""
    fb = [DeviceBuffer.from_numpy(np.ascontiguousarray(f, np.float16)) for f in feats]
    N, K = rois.shape[:2]
    Cc = feats[0].shape[3]
    ptrs = (C.c_void_p * len(fb))(*[b.ptr.value for b in fb])
    Hs = (C.c_int32 * len(fb))(*[f.shape[1] for f in feats]); Ws = (C.c_int32 * len(fb))(*[f.shape[2] for f in feats])
    sc = (C.c_float * len(fb))(*scales)
    dr = DeviceBuffer.from_numpy(np.ascontiguousarray(rois, np.float32)); dcnt = DeviceBuffer.from_numpy(np.ascontiguousarray(counts, np.int32))
    do = DeviceBuffer((N * K, PH, PW, Cc), np.float16)
    check(lib().isegmi_op_roi_align_f16(ptrs, Hs, Ws, sc, len(fb), dr.ptr, dcnt.ptr, N, K, Cc, PH, PW, sampling, aligned, k_min, do.ptr, None))
    return do.numpy()


class BoxPostArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "R", "ncls", "det_per_img", "cap", "nms_flags")] + \
               [("score_thresh", C.c_float), ("nms_thresh", C.c_float), ("logits_stride", C.c_int64), ("regr_stride", C.c_int64)] + \
               [(n, C.c_void_p) for n in ("d_logits", "d_regr", "d_props", "d_prop_cnt", "d_image_hw", "d_ws_prob", "d_ws_cand_scores",
                                          "d_ws_cand_boxes", "d_ws_kept_total", "d_ws_top_vals", "d_ws_top_idx", "d_out_count",
                                          "d_out_boxes", "d_out_scores", "d_out_labels", "d_ws_crowd_matrix", "d_ws_crowd_keys",
                                          "d_ws_crowd_boxes", "d_ws_crowd_m")]


def box_postprocess(logits, regr, props, prop_cnt, image_hw, score_thr=0.05, nms_thr=0.5, det_per_img=100, nms_flags=0, cap=0, chip_wide=True):
    """logits [N,R,ncls], regr [N,R,4*ncls], props [N,R,4] -> list of (boxes, scores, labels).  cap > det_per_img: rows for the detections
    that tie with the det_per_img-th score (upstream's kth-value rule keeps them).  chip_wide: hand the op its optional crowd workspaces (classes
    with more than 128 candidates get their suppression matrix from the whole chip: three launches); False = every class in its own block."""
    logits = np.ascontiguousarray(logits, np.float32)
    N, R, ncls = logits.shape
    cap = max(det_per_img, cap)
    b = dict(d_logits=DeviceBuffer.from_numpy(logits), d_regr=DeviceBuffer.from_numpy(np.ascontiguousarray(regr, np.float32)),
             d_props=DeviceBuffer.from_numpy(np.ascontiguousarray(props, np.float32)),
             d_prop_cnt=DeviceBuffer.from_numpy(np.ascontiguousarray(prop_cnt, np.int32)),
             d_image_hw=DeviceBuffer.from_numpy(np.ascontiguousarray(image_hw, np.int32)),
             d_ws_prob=DeviceBuffer((N, R, ncls)), d_ws_cand_scores=DeviceBuffer((N, ncls - 1, R)),
             d_ws_cand_boxes=DeviceBuffer((N, ncls - 1, R, 4)), d_ws_kept_total=DeviceBuffer((N,), np.int32),
             d_ws_top_vals=DeviceBuffer((N, det_per_img)), d_ws_top_idx=DeviceBuffer((N, det_per_img), np.int32),
             d_out_count=DeviceBuffer((N,), np.int32), d_out_boxes=DeviceBuffer((N, cap, 4)), d_out_scores=DeviceBuffer((N, cap)),
             d_out_labels=DeviceBuffer((N, cap), np.int32))
    if chip_wide:   # dirty on purpose: nothing may be read before it is written
        b.update(d_ws_crowd_matrix=DeviceBuffer.from_numpy(np.full((N, ncls - 1, 131072), 0xa5, np.uint8)), d_ws_crowd_keys=DeviceBuffer((N, ncls - 1, R), np.int64),
                 d_ws_crowd_boxes=DeviceBuffer((N, ncls - 1, R, 4)), d_ws_crowd_m=DeviceBuffer.from_numpy(np.full((N, ncls - 1), 777, np.int32)))
    a = BoxPostArgs(N, R, ncls, det_per_img, cap, nms_flags, score_thr, nms_thr, ncls, 4 * ncls, *[b[n].ptr if n in b else None for n, _ in BoxPostArgs._fields_[10:]])
    check(lib().isegmi_op_box_postprocess(C.byref(a), None))
    cnt = b["d_out_count"].numpy(); B = b["d_out_boxes"].numpy(); S = b["d_out_scores"].numpy(); Lb = b["d_out_labels"].numpy()
    return [(B[i, : cnt[i]], S[i, : cnt[i]], Lb[i, : cnt[i]]) for i in range(N)]


def mask_logits_select(feat, w, b, labels):
    feat = np.ascontiguousarray(feat, np.float32)
    R, HW, Cc = feat.shape
    df = DeviceBuffer.from_numpy(feat); dw = DeviceBuffer.from_numpy(np.ascontiguousarray(w, np.float32))
    db = DeviceBuffer.from_numpy(np.ascontiguousarray(b, np.float32)); dl = DeviceBuffer.from_numpy(np.ascontiguousarray(labels, np.int32))
    do = DeviceBuffer((R, HW))
    check(lib().isegmi_op_mask_logits_select(df.ptr, R, HW, Cc, dw.ptr, db.ptr, dl.ptr, do.ptr, None))
    return do.numpy()


def paste_masks(masks, boxes, counts, im_h, im_w, thr=0.5):
    masks = np.ascontiguousarray(masks, np.float32)
    N, K, M, _ = masks.shape
    dm = DeviceBuffer.from_numpy(masks); db = DeviceBuffer.from_numpy(np.ascontiguousarray(boxes, np.float32))
    dc = DeviceBuffer.from_numpy(np.ascontiguousarray(counts, np.int32)); do = DeviceBuffer((N, K, im_h, im_w), np.uint8)
    check(lib().isegmi_op_paste_masks(dm.ptr, db.ptr, dc.ptr, N, K, M, im_h, im_w, C.c_float(thr), do.ptr, None))
    return do.numpy()


def rpn_level(head, anchors, image_hw, A, pre_nms, post_nms, nms_thr=0.7, min_size=0.0, nms_flags=0, chip_wide=True):
    """head [N,H,W,A*5] fused (A logits, A*4 deltas) -> list of (boxes, scores).  chip_wide: hand the op its optional
    suppression-matrix workspace (two-kernel NMS, pre_nms <= 1024); False = single-block NMS."""
    head = np.ascontiguousarray(head, np.float32)
    N, H, W, CH = head.shape
    HW = H * W
    dh = DeviceBuffer.from_numpy(head); da = DeviceBuffer.from_numpy(np.ascontiguousarray(anchors, np.float32))
    dhw = DeviceBuffer.from_numpy(np.ascontiguousarray(image_hw, np.int32))
    wp = DeviceBuffer((N, HW * A)); tv = DeviceBuffer((N, pre_nms)); ti = DeviceBuffer((N, pre_nms), np.int32); tc = DeviceBuffer((N,), np.int32)
    ob = DeviceBuffer((N, post_nms, 4)); os_ = DeviceBuffer((N, post_nms)); oc = DeviceBuffer((N,), np.int32)
    wn = DeviceBuffer((N, 131072), np.uint8) if chip_wide and pre_nms <= 1024 else None
    check(lib().isegmi_op_rpn_level(dh.ptr, da.ptr, dhw.ptr, N, HW, A, pre_nms, post_nms, C.c_float(nms_thr), C.c_float(min_size), nms_flags,
                                    wp.ptr, tv.ptr, ti.ptr, tc.ptr, ob.ptr, os_.ptr, oc.ptr, wn.ptr if wn is not None else None, None))
    c = oc.numpy(); B = ob.numpy(); S = os_.numpy()
    return [(B[i, : c[i]], S[i, : c[i]]) for i in range(N)]


def rpn_levels(heads, anchors, image_hw, A, pre_nms, post_nms, nms_thr=0.7, min_size=0.0, nms_flags=0):
    """All FPN levels at once (isegmi_op_rpn_levels): heads[l] [N,H_l,W_l,A*5], anchors[l] [H_l*W_l*A,4] -> per level a list over images of (boxes, scores)."""
    nl = len(heads)
    hs = [np.ascontiguousarray(h, np.float32) for h in heads]
    N = hs[0].shape[0]
    HW = (C.c_int32 * nl)(*[h.shape[1] * h.shape[2] for h in hs])
    dh = [DeviceBuffer.from_numpy(h) for h in hs]; da = [DeviceBuffer.from_numpy(np.ascontiguousarray(a, np.float32)) for a in anchors]
    ph = (C.c_void_p * nl)(*[b.ptr.value for b in dh]); pa = (C.c_void_p * nl)(*[b.ptr.value for b in da])
    dhw = DeviceBuffer.from_numpy(np.ascontiguousarray(image_hw, np.int32))
    pe, ce = C.c_int64(), C.c_int64()
    check(lib().isegmi_op_rpn_levels_workspace(nl, N, HW, A, pre_nms, C.byref(pe), C.byref(ce)))
    wp = DeviceBuffer((pe.value,)); cv = DeviceBuffer((ce.value,)); ci = DeviceBuffer((ce.value,), np.int32)
    tv = DeviceBuffer((nl, N, pre_nms)); ti = DeviceBuffer((nl, N, pre_nms), np.int32); tc = DeviceBuffer((nl, N), np.int32)
    wn = DeviceBuffer((nl * N, 131072), np.uint8)
    ob = DeviceBuffer((N, nl, post_nms, 4)); os_ = DeviceBuffer((N, nl, post_nms)); oc = DeviceBuffer((N, nl), np.int32)
    check(lib().isegmi_op_rpn_levels(nl, ph, pa, HW, dhw.ptr, N, A, pre_nms, post_nms, C.c_float(nms_thr), C.c_float(min_size), nms_flags, wp.ptr, cv.ptr, ci.ptr,
                                     tv.ptr, ti.ptr, tc.ptr, wn.ptr, ob.ptr, os_.ptr, oc.ptr, None))
    c = oc.numpy(); B = ob.numpy(); S = os_.numpy()
    return [[(B[i, l, : c[i, l]], S[i, l, : c[i, l]]) for i in range(N)] for l in range(nl)]


# ---------------------------------------------------------------- fp16 conv (configs[4])
def pack_conv_weights_f16(desc, w_krsc):
    w = np.ascontiguousarray(w_krsc, np.float32)
    n = C.c_int64()
    check(lib().isegmi_conv_packed_halfs(C.byref(desc), C.byref(n)))
    out = np.empty(n.value, np.uint16)
    check(lib().isegmi_pack_conv_weights_f16(C.byref(desc), w.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    return out.view(np.float16)


def stem_f16(x_nhwc3, w_krs4, scale=None, shift=None, tile=0):
    """fp16 stem: fp32 NHWC C=3 batch -> (haloed fp16 image) -> 7x7/2 conv + scale/shift + ReLU, fp16 out."""
    x = np.ascontiguousarray(x_nhwc3, np.float32)
    N, H, W, _ = x.shape
    Cout = w_krs4.shape[0]
    d = make_conv_desc(N, H, W, 4, Cout, 7, 7, 2, 3, 1, tile)
    dx = DeviceBuffer.from_numpy(x); dh = DeviceBuffer((N, H + 6, (W + 7) & ~1, 4), np.float16)
    check(lib().isegmi_op_pad_c3_to_f16_halo(dx.ptr, N, H, W, dh.ptr, None))
    dw = DeviceBuffer.from_numpy(pack_conv_weights_f16(d, w_krs4))
    ds = None if scale is None else DeviceBuffer.from_numpy(np.asarray(scale, np.float32))
    dsh = None if shift is None else DeviceBuffer.from_numpy(np.asarray(shift, np.float32))
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    do = DeviceBuffer((N, ho, wo, Cout), np.float16)
    check(lib().isegmi_op_conv2d_f16(C.byref(d), dh.ptr, dw.ptr, _ptr(ds), _ptr(dsh), None, do.ptr, 0, None))
    return do.numpy(), dh.numpy()


def conv1x1_up2x_add_f16(x, w, scale, shift, coarse):
    """isegmi_op_conv1x1_up2x_add_f16: x fp16 NHWC, w [Cout,1,1,Cin], coarse fp16 [N,Hc,Wc,Cout] -> fp16 [N,H,W,Cout]."""
    x = np.ascontiguousarray(x, np.float16); coarse = np.ascontiguousarray(coarse, np.float16)
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    d = make_conv_desc(N, H, W, Cin, Cout, 1, 1, 1, 0, 0, 0)
    dx = DeviceBuffer.from_numpy(x); dw = DeviceBuffer.from_numpy(pack_conv_weights_f16(d, w)); dc = DeviceBuffer.from_numpy(coarse)
    ds = DeviceBuffer.from_numpy(np.asarray(scale, np.float32)); dsh = DeviceBuffer.from_numpy(np.asarray(shift, np.float32))
    do = DeviceBuffer((N, H, W, Cout), np.float16)
    check(lib().isegmi_op_conv1x1_up2x_add_f16(C.byref(d), dx.ptr, dw.ptr, ds.ptr, dsh.ptr, dc.ptr, coarse.shape[1], coarse.shape[2], do.ptr, None))
    return do.numpy()


def conv3x3_head_f16(x, w, scale, shift, w2, scale2, shift2):
    """isegmi_op_conv3x3_head_f16: x fp16 NHWC [N,H,W,Cin]; w [256,3,3,Cin]; w2 [cout2,1,1,256].  Returns (fp32 [N,H,W,cout2] or None, fused flag)."""
    x = np.ascontiguousarray(x, np.float16)
    N, H, W, Cin = x.shape
    cout2 = w2.shape[0]
    d = make_conv_desc(N, H, W, Cin, 256, 3, 3, 1, 1, 1, 0)
    d2 = make_conv_desc(N, H, W, 256, cout2, 1, 1, 1, 0, 0, 0)
    dx = DeviceBuffer.from_numpy(x); dw = DeviceBuffer.from_numpy(pack_conv_weights_f16(d, w)); dw2 = DeviceBuffer.from_numpy(pack_conv_weights_f16(d2, w2))
    bufs = [DeviceBuffer.from_numpy(np.asarray(a, np.float32)) for a in (scale, shift, scale2, shift2)]
    do = DeviceBuffer((N, H, W, cout2), np.float32)
    fused = C.c_int(0)
    check(lib().isegmi_op_conv3x3_head_f16(C.byref(d), dx.ptr, dw.ptr, bufs[0].ptr, bufs[1].ptr, dw2.ptr, bufs[2].ptr, bufs[3].ptr, cout2, do.ptr, C.byref(fused), None))
    return (do.numpy() if fused.value else None), bool(fused.value)


def stem_pool_f16(x_nhwc3, w_krs4, scale, shift, flags=0):
    """The fp16 stem in one launch (isegmi_op_stem_pool_f16): fp32 NHWC C=3 batch -> haloed fp16 image -> conv 7x7/2 + BN + ReLU + max-pool 3x3/2, fp16 out."""
    x = np.ascontiguousarray(x_nhwc3, np.float32)
    N, H, W, _ = x.shape
    Cout = w_krs4.shape[0]
    d = make_conv_desc(N, H, W, 4, Cout, 7, 7, 2, 3, 1, 0)
    dx = DeviceBuffer.from_numpy(x); dh = DeviceBuffer((N, H + 6, (W + 7) & ~1, 4), np.float16)
    check(lib().isegmi_op_pad_c3_to_f16_halo(dx.ptr, N, H, W, dh.ptr, None))
    dw = DeviceBuffer.from_numpy(pack_conv_weights_f16(d, w_krs4))
    ds = DeviceBuffer.from_numpy(np.asarray(scale, np.float32)); dsh = DeviceBuffer.from_numpy(np.asarray(shift, np.float32))
    hc, wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    do = DeviceBuffer((N, (hc - 1) // 2 + 1, (wc - 1) // 2 + 1, Cout), np.float16)
    check(lib().isegmi_op_stem_pool_f16(N, H, W, dh.ptr, dw.ptr, ds.ptr, dsh.ptr, do.ptr, int(flags), None))
    return do.numpy()


def conv2d_f16(x, w_krsc, stride=1, pad=0, scale=None, shift=None, residual=None, act=0, tile=0, out_f32=False):
    """x fp16-representable NHWC (any float dtype; cast to fp16), returns fp16 (or fp32 when out_f32) as numpy."""
    x = np.ascontiguousarray(x, np.float16)
    N, H, W, Cin = x.shape
    Cout, R, S, _ = w_krsc.shape
    d = make_conv_desc(N, H, W, Cin, Cout, R, S, stride, pad, act, tile)
    ho, wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    dx = DeviceBuffer.from_numpy(x); dw = DeviceBuffer.from_numpy(pack_conv_weights_f16(d, w_krsc))
    ds = None if scale is None else DeviceBuffer.from_numpy(np.asarray(scale, np.float32))
    dh = None if shift is None else DeviceBuffer.from_numpy(np.asarray(shift, np.float32))
    dr = None if residual is None else DeviceBuffer.from_numpy(np.ascontiguousarray(residual, np.float16))
    do = DeviceBuffer((N, ho, wo, Cout), np.float32 if out_f32 else np.float16)
    check(lib().isegmi_op_conv2d_f16(C.byref(d), dx.ptr, dw.ptr, _ptr(ds), _ptr(dh), _ptr(dr), do.ptr, int(out_f32), None))
    return do.numpy()


class BottleneckDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "H", "W", "Cin", "Cmid", "flags")]


def bottleneck_ds_f16(x, w1, sb1, w2, sb2, w3, sb3, wd, sbd, flags=0):
    """First block of res2 with its projection shortcut (isegmi_op_bottleneck_ds_f16): x fp16 [N,H,W,64]; wd [256,1,1,64].  Returns fp16 [N,H,W,256]."""
    x = np.ascontiguousarray(x, np.float16)
    N, H, W, Cin = x.shape
    Cmid = w1.shape[0]
    bufs = []
    for w, (sc, sh) in ((w1, sb1), (w2, sb2), (w3, sb3), (wd, sbd)):
        Cout, R, S, Ci = w.shape
        d = make_conv_desc(N, H, W, Ci, Cout, R, S, 1, R // 2, 1, 0)
        bufs += [DeviceBuffer.from_numpy(pack_conv_weights_f16(d, w)), DeviceBuffer.from_numpy(np.asarray(sc, np.float32)),
                 DeviceBuffer.from_numpy(np.asarray(sh, np.float32))]
    dx = DeviceBuffer.from_numpy(x)
    do = DeviceBuffer((N, H, W, 4 * Cmid), np.float16)
    bd = BottleneckDesc(N, H, W, Cin, Cmid, int(flags))
    check(lib().isegmi_op_bottleneck_ds_f16(C.byref(bd), dx.ptr, *[b.ptr for b in bufs], do.ptr, None))
    return do.numpy()


def bottleneck_f16(x, w1, sb1, w2, sb2, w3, sb3, flags=0):
    """Fused identity bottleneck (isegmi_op_bottleneck_f16): x fp16 NHWC [N,H,W,Cin]; w1 [Cmid,1,1,Cin], w2 [Cmid,3,3,Cmid],
    w3 [Cin,1,1,Cmid] natural KRSC fp32 (rounded to fp16 by the packer); sb* = (scale, shift) of the folded BN.  Returns fp16 numpy."""
    x = np.ascontiguousarray(x, np.float16)
    N, H, W, Cin = x.shape
    Cmid = w1.shape[0]
    bufs = []
    for w, (sc, sh) in ((w1, sb1), (w2, sb2), (w3, sb3)):
        Cout, R, S, Ci = w.shape
        d = make_conv_desc(N, H, W, Ci, Cout, R, S, 1, R // 2, 1, 0)
        bufs += [DeviceBuffer.from_numpy(pack_conv_weights_f16(d, w)), DeviceBuffer.from_numpy(np.asarray(sc, np.float32)),
                 DeviceBuffer.from_numpy(np.asarray(sh, np.float32))]
    dx = DeviceBuffer.from_numpy(x)
    do = DeviceBuffer((N, H, W, Cin), np.float16)
    bd = BottleneckDesc(N, H, W, Cin, Cmid, int(flags))
    check(lib().isegmi_op_bottleneck_f16(C.byref(bd), dx.ptr, *[b.ptr for b in bufs], do.ptr, None))
    return do.numpy()


# ---------------------------------------------------------------- COCO RLE on the device
class RleArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("N", "K", "plane_h", "plane_w", "cap_runs", "cap_chars")] + \
               [(n, C.c_void_p) for n in ("d_masks", "d_count", "d_image_hw", "d_windows", "d_ws_trans", "d_ws_col", "d_ws_nruns", "d_ws_tile", "d_ws_len",
                                          "d_ws_starts", "d_out_run_off", "d_out_counts", "d_out_str_off", "d_out_chars", "d_out_status")]


def rle_encode(masks, count=None, image_hw=None, cap_runs=None, cap_chars=None, windows=None):
    """masks [N,K,h,w] uint8 -> (run_off [N*K+1], counts, str_off [N*K+1], chars bytes, status [4]): pycocotools rleEncode + rleToString of
    every valid slot (k < count[n]) over the top-left image_hw[n] window of its plane (host convenience wrapper of isegmi_op_rle_encode).
    windows [N, K, 4] (x0, y0, x1, y1): every set pixel of a slot lies inside its window; nothing outside it is read."""
    masks = np.ascontiguousarray(masks, np.uint8)
    N, K, h, w = masks.shape
    cap_runs = int(cap_runs or max(1024, -(-(masks.size + N * K) // 1024) * 1024))  # worst case: every pixel starts a run
    cap_chars = int(cap_chars or 7 * cap_runs)
    sz = [C.c_int64() for _ in range(6)]
    check(lib().isegmi_rle_workspace(N, K, h, w, cap_runs, *[C.byref(s) for s in sz]))
    ws = [DeviceBuffer((s.value,), np.uint8) for s in sz]
    dm = DeviceBuffer.from_numpy(masks)
    dc = None if count is None else DeviceBuffer.from_numpy(np.ascontiguousarray(count, np.int32))
    dh = None if image_hw is None else DeviceBuffer.from_numpy(np.ascontiguousarray(image_hw, np.int32).reshape(N, 2))
    dwin = None if windows is None else DeviceBuffer.from_numpy(np.ascontiguousarray(windows, np.int32).reshape(N, K, 4))
    ro = DeviceBuffer((N * K + 1,), np.int32); cn = DeviceBuffer((cap_runs,), np.uint32); so = DeviceBuffer((N * K + 1,), np.int32)
    ch = DeviceBuffer((cap_chars,), np.uint8); stt = DeviceBuffer((4,), np.int32)
    a = RleArgs(N, K, h, w, cap_runs, cap_chars, dm.ptr, _ptr(dc), _ptr(dh), _ptr(dwin), ws[0].ptr, ws[1].ptr, ws[2].ptr, ws[3].ptr, ws[4].ptr, ws[5].ptr,
                ro.ptr, cn.ptr, so.ptr, ch.ptr, stt.ptr)
    check(lib().isegmi_op_rle_encode(C.byref(a), None))
    sync()
    return ro.numpy(), cn.numpy(), so.numpy(), ch.numpy().tobytes(), stt.numpy()
