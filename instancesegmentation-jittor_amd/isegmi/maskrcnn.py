"""Mask R-CNN R50/R101-FPN inference, host side (mirrors detectron.jittor's predictor surface).

Reference surface (README.md:288-335): `cfg.merge_from_file(...)`; `COCODemo(cfg, min_image_size=800,
confidence_threshold=0.5)`; `coco_demo.run_on_opencv_image(image)`; [UPSTREAM-RECALL, SURVEY 8b]
`compute_prediction(image) -> BoxList` with fields scores / labels / mask.  All numerics run in
libisegmi.so; this module moves config, weights, anchors and results across the C ABI.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _ffi
from .weights import fold_frozen_batchnorm, to_krsc

PIXEL_MEAN = (102.9801, 115.9465, 122.7717)  # BGR, TO_BGR255


@dataclass(frozen=True)
class MaskRCNNConfig:
    """e2e_mask_rcnn_R_50_FPN_1x inference constants; key names follow the yaml."""
    depth: int = 50
    MIN_SIZE_TEST: int = 800
    MAX_SIZE_TEST: int = 1333
    SIZE_DIVISIBILITY: int = 32
    ANCHOR_SIZES: tuple = (32, 64, 128, 256, 512)
    ANCHOR_STRIDE: tuple = (4, 8, 16, 32, 64)
    ASPECT_RATIOS: tuple = (0.5, 1.0, 2.0)
    RPN_PRE_NMS_TOP_N_TEST: int = 1000
    RPN_POST_NMS_TOP_N_TEST: int = 1000
    RPN_FPN_POST_NMS_TOP_N_TEST: int = 1000
    RPN_NMS_THRESH: float = 0.7
    RPN_MIN_SIZE: float = 0.0
    ROI_SCORE_THRESH: float = 0.05
    ROI_NMS: float = 0.5
    DETECTIONS_PER_IMG: int = 100
    DETECTIONS_CAP: int = 0  # rows per image in the detection buffers; 0 = DETECTIONS_PER_IMG.  upstream's kth-value cut keeps every detection that
    #                          TIES with the 100th score, so an image can return more than DETECTIONS_PER_IMG: set e.g. 128 to receive those ties
    #                          (the mask head then runs over that many RoI slots per image); with 0 ties past the 100th row are cut in output order
    # Semantic forks: which side detectron.jittor takes cannot be checked (the reference tree holds no code), so each
    # is a switch, mirrored in the oracle (oracle/maskrcnn_ref.py) and run through the engine by tests/test_forks_gpu.py.  Defaults = maskrcnn-benchmark's
    # CUDA path.
    NMS_GE: int = 0             # 0 suppress on iou > thr (CUDA kernel, jt.nms), 1 on >= (CPU loop)
    NMS_PLUS_ONE: int = 1       # 1 legacy +1 widths in the NMS IoU (RPN and box post-processing), 0 plain areas
    NMS_OUTPUT_ORDER: str = "score"   # order of a class's detections after the box NMS: "score" (CUDA kernel) or "index" (CPU: nonzero(keep), proposal order)
    ROI_ALIGNED: int = 0        # 0 legacy ROIAlign (no half-pixel shift, RoI >= 1 pixel), 1 ROIAlign(aligned=True)
    FROZEN_BN_EPS: float = 0.0  # FrozenBatchNorm2d: scale = w * rsqrt(var + eps); maskrcnn-benchmark has no eps
    # Opt-in NUMERICS MODE for latency (not a semantic fork: the same sums in another fp32 association): the backbone's small-M / large-K bottleneck
    # convolutions (a bs = 1 forward) as four k-ordered partial chains added left to right -- conv tile 15, include/isegmi.h; bit-exact against the oracle
    # model run with conv_split_k=1, NOT against the default mode; results then depend on the batch size a layer is run at (the rule is by shape)
    CONV_SPLIT_K: int = 0
    CONV_BODY: str = "R-50-FPN"  # "R-50-FPN" / "R-101-FPN" (depth) or "R-50-C4" (the yaml README.md:263-273 prints)

    @staticmethod
    def c4():
        """e2e_mask_rcnn_R_50_C4_1x: one stride-16 map, 15 anchors, PRE/POST_NMS_TOP_N_TEST 6000/1000 (README.md:267-269),
        ROIAlign 14x14 with sampling_ratio 0, conv5 head shared by box and mask branches, MaskRCNNC4Predictor (14x14 masks)."""
        return MaskRCNNConfig(CONV_BODY="R-50-C4", SIZE_DIVISIBILITY=16, ANCHOR_STRIDE=(16,), RPN_PRE_NMS_TOP_N_TEST=6000,
                              RPN_POST_NMS_TOP_N_TEST=1000)

    @property
    def is_c4(self):
        return self.CONV_BODY.endswith("-C4")

    @property
    def det_cap(self):
        return max(self.DETECTIONS_PER_IMG, self.DETECTIONS_CAP)


# ---------------------------------------------------------------------------------------- anchors (A.3)
def _whctrs(a):
    w = a[2] - a[0] + 1
    h = a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mkanchors(ws, hs, xc, yc):
    ws = ws[:, None]; hs = hs[:, None]
    return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def generate_anchors(stride, size, aspect_ratios):
    anchor = np.array([1, 1, stride, stride], np.float64) - 1
    w, h, xc, yc = _whctrs(anchor)
    ratios = np.asarray(aspect_ratios, np.float64)
    ws = np.round(np.sqrt(w * h / ratios)); hs = np.round(ws * ratios)
    ra = _mkanchors(ws, hs, xc, yc)
    scales = np.array([size / stride], np.float64)
    out = []
    for i in range(ra.shape[0]):
        w, h, xc, yc = _whctrs(ra[i])
        out.append(_mkanchors(w * scales, h * scales, xc, yc))
    return np.vstack(out).astype(np.float32)


def generate_anchors_multi(stride, sizes, aspect_ratios):
    """Single-feature-map form (R-50-C4: stride 16, sizes 32..512): ratio enumeration outside, ALL scales inside, i.e.
    anchor index = ratio * len(sizes) + size (maskrcnn-benchmark generate_anchors with several sizes)."""
    anchor = np.array([1, 1, stride, stride], np.float64) - 1
    w, h, xc, yc = _whctrs(anchor)
    ratios = np.asarray(aspect_ratios, np.float64)
    ws = np.round(np.sqrt(w * h / ratios)); hs = np.round(ws * ratios)
    ra = _mkanchors(ws, hs, xc, yc)
    scales = np.asarray(sizes, np.float64) / stride
    out = []
    for i in range(ra.shape[0]):
        w, h, xc, yc = _whctrs(ra[i])
        out.append(_mkanchors(w * scales, h * scales, xc, yc))
    return np.vstack(out).astype(np.float32)


def grid_anchors(grid_h, grid_w, stride, base):
    sx = np.arange(0, grid_w * stride, stride, dtype=np.float32)
    sy = np.arange(0, grid_h * stride, stride, dtype=np.float32)
    yy, xx = np.meshgrid(sy, sx, indexing="ij")
    shifts = np.stack([xx.ravel(), yy.ravel(), xx.ravel(), yy.ravel()], 1)
    return (shifts[:, None, :] + base[None, :, :]).reshape(-1, 4).astype(np.float32)


def level_shapes(H, W):
    s = ((H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1)
    s = ((s[0] + 2 - 3) // 2 + 1, (s[1] + 2 - 3) // 2 + 1)  # C2 (stride 4)
    out = [s]
    for _ in range(3):
        s = ((s[0] - 1) // 2 + 1, (s[1] - 1) // 2 + 1)      # 1x1 stride 2
        out.append(s)
    out.append(((s[0] - 1) // 2 + 1, (s[1] - 1) // 2 + 1))  # P6
    return out


def prepare_images(images_bgr_f32, divisibility=32):
    """to_image_list (M1) for already-resized BGR 0..255 float images: subtract PIXEL_MEAN, zero-pad to a
    common size divisible by 32.  Returns (batch NHWC3 fp32, image_hw [N,2] int32)."""
    hw = np.array([im.shape[:2] for im in images_bgr_f32], np.int32)
    H = int(-(-hw[:, 0].max() // divisibility) * divisibility)
    W = int(-(-hw[:, 1].max() // divisibility) * divisibility)
    out = np.zeros((len(images_bgr_f32), H, W, 3), np.float32)
    mean = np.asarray(PIXEL_MEAN, np.float32)
    for i, im in enumerate(images_bgr_f32):
        out[i, : im.shape[0], : im.shape[1]] = np.asarray(im, np.float32) - mean
    return out, hw


class BoxList:
    """Minimal structures/bounding_box.BoxList (M13): xyxy boxes, size=(w,h), named fields."""

    def __init__(self, bbox, image_size, mode="xyxy"):
        self.bbox = np.asarray(bbox, np.float32).reshape(-1, 4)
        self.size = tuple(image_size)
        self.mode = mode
        self.extra_fields = {}

    def add_field(self, k, v):
        self.extra_fields[k] = v

    def get_field(self, k):
        return self.extra_fields[k]

    def has_field(self, k):
        return k in self.extra_fields

    def fields(self):
        return list(self.extra_fields)

    def __len__(self):
        return self.bbox.shape[0]

    def __getitem__(self, item):
        b = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            b.add_field(k, v[item])
        return b

    def resize(self, size):
        rw, rh = (float(s) / float(o) for s, o in zip(size, self.size))
        r = np.array([rw, rh, rw, rh], np.float32)
        b = BoxList(self.bbox * r, size, self.mode)
        for k, v in self.extra_fields.items():
            b.add_field(k, v)
        return b


def padded_canvas(image_hw, divisibility=32):
    """to_image_list: the canvas of a batch = max height / width over its images, rounded up to SIZE_DIVISIBILITY."""
    hw = np.asarray(image_hw, np.int64).reshape(-1, 2)
    return (int(-(-hw[:, 0].max() // divisibility) * divisibility), int(-(-hw[:, 1].max() // divisibility) * divisibility))


class MaskRCNN:
    """GeneralizedRCNN engine wrapper: `model = MaskRCNN(sd, H, W)`; `preds = model(batch, image_hw)`.

    (H, W) is the LARGEST padded canvas the engine serves.  Every batch runs on its own canvas (to_image_list pads a batch to the maximum
    size of its images, rounded up to SIZE_DIVISIBILITY; the RPN sees the padding, so results depend on the canvas exactly as upstream's
    do): weights are packed once, anchors are laid out on the device per canvas, and `reserve()` sizes every buffer once."""

    KIND = 2

    def __init__(self, state_dict, H, W, cfg=MaskRCNNConfig(), max_batch=2, device=0, fp16=False):
        assert H % cfg.SIZE_DIVISIBILITY == 0 and W % cfg.SIZE_DIVISIBILITY == 0
        assert not (cfg.is_c4 and fp16), "the C4 configuration is fp32 only"
        self.cfg, self.H, self.W, self.max_batch = cfg, H, W, max_batch
        self.mask_buf = "det.mask14" if cfg.is_c4 else "det.mask28"
        _ffi.lib()
        _ffi.set_device(device)
        self._h = C.c_void_p()
        _ffi.check(_ffi.lib().isegmi_engine_create(self.KIND, max_batch, H, W, C.byref(self._h)))
        self.fp16 = bool(fp16)
        if self.fp16:  # fp16 storage + f16 MFMA convs (BASELINE configs[4]); must precede weight loading
            self.set_param("fp16", 1.0)
        if cfg.is_c4:
            self.set_param("arch_c4", 1.0)
            self._load_c4(state_dict)
        else:
            self._load(state_dict)
        for k, v in (("resnet_depth", cfg.depth), ("rpn_pre_nms_top_n", cfg.RPN_PRE_NMS_TOP_N_TEST),
                     ("rpn_post_nms_top_n", cfg.RPN_POST_NMS_TOP_N_TEST), ("rpn_fpn_post_nms_top_n", cfg.RPN_FPN_POST_NMS_TOP_N_TEST),
                     ("rpn_nms_thresh", cfg.RPN_NMS_THRESH), ("rpn_min_size", cfg.RPN_MIN_SIZE), ("roi_score_thresh", cfg.ROI_SCORE_THRESH),
                     ("roi_nms_thresh", cfg.ROI_NMS), ("detections_per_img", cfg.DETECTIONS_PER_IMG), ("detections_cap", cfg.DETECTIONS_CAP),
                     ("nms_ge", cfg.NMS_GE), ("nms_plus_one", cfg.NMS_PLUS_ONE), ("nms_index_order", {"score": 0, "index": 1}[cfg.NMS_OUTPUT_ORDER]),
                     ("roi_aligned", cfg.ROI_ALIGNED), ("conv_split_k", cfg.CONV_SPLIT_K)):
            self.set_param(k, float(v))
        self._d_in = _ffi.DeviceBuffer((max_batch, H, W, 3))
        self._hw = None
        self._canvas = (H, W)

    def reserve(self):
        """Size every activation / workspace / output buffer for the largest canvas and batch by running one forward + paste on a zero batch
        (buffers only ever grow, so no forward re-allocates afterwards).  Returns memory()."""
        self._d_in.zero()
        self._hw = np.tile(np.array([[self.H, self.W]], np.int32), (self.max_batch, 1))
        self._canvas = (self.H, self.W)
        self.forward_device(self.max_batch)
        self.paste_device(self.H, self.W)
        self.sync()
        return self.memory()


    def set_param(self, name, value):
        _ffi.check(_ffi.lib().isegmi_engine_set_param(self._h, name.encode(), C.c_float(value)))

    def _set_conv_krsc(self, name, w, scale=None, shift=None):
        w = np.ascontiguousarray(w, np.float32)
        cout, r, s, cin = w.shape
        fp = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).ctypes.data_as(C.c_void_p)
        sc = None if scale is None else np.ascontiguousarray(scale, np.float32)
        sh = None if shift is None else np.ascontiguousarray(shift, np.float32)
        _ffi.check(_ffi.lib().isegmi_engine_set_conv(self._h, name.encode(), cout, r, s, cin, fp(w), fp(sc), fp(sh)))

    def _set_tensor(self, name, a):
        a = np.ascontiguousarray(a)
        _ffi.check(_ffi.lib().isegmi_engine_set_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), C.c_int64(a.nbytes)))

    def _load(self, sd):
        cfg = self.cfg
        w = to_krsc(sd["backbone.body.stem.conv1.weight"])
        w = np.concatenate([w, np.zeros(w.shape[:3] + (1,), np.float32)], -1)
        self._set_conv_krsc("backbone.body.stem.conv1", w, *fold_frozen_batchnorm(sd, "backbone.body.stem.bn1", cfg.FROZEN_BN_EPS))
        blocks = (3, 4, 23 if cfg.depth == 101 else 6, 3)
        for li, nb in enumerate(blocks, 1):
            for b in range(nb):
                nm = "backbone.body.layer%d.%d" % (li, b)
                for i in (1, 2, 3):
                    self._set_conv_krsc("%s.conv%d" % (nm, i), to_krsc(sd["%s.conv%d.weight" % (nm, i)]),
                                        *fold_frozen_batchnorm(sd, "%s.bn%d" % (nm, i), cfg.FROZEN_BN_EPS))
                if b == 0:
                    self._set_conv_krsc(nm + ".downsample.0", to_krsc(sd[nm + ".downsample.0.weight"]),
                                        *fold_frozen_batchnorm(sd, nm + ".downsample.1", cfg.FROZEN_BN_EPS))
        for i in range(1, 5):
            for k in ("inner", "layer"):
                nm = "backbone.fpn.fpn_%s%d" % (k, i)
                self._set_conv_krsc(nm, to_krsc(sd[nm + ".weight"]), None, sd[nm + ".bias"])
        self._set_conv_krsc("rpn.head.conv", to_krsc(sd["rpn.head.conv.weight"]), None, sd["rpn.head.conv.bias"])
        # fused 1x1: A objectness logits followed by A*4 deltas
        wcb = np.concatenate([to_krsc(sd["rpn.head.cls_logits.weight"]), to_krsc(sd["rpn.head.bbox_pred.weight"])], 0)
        bcb = np.concatenate([sd["rpn.head.cls_logits.bias"], sd["rpn.head.bbox_pred.bias"]])
        self._set_conv_krsc("rpn.head.cls_bbox", wcb, None, bcb)
        # FC6 consumes NHWC RoI features: permute its (c,h,w) input ordering to (h,w,c) -> a 7x7 'valid' conv
        w6 = sd["roi_heads.box.feature_extractor.fc6.weight"].reshape(1024, 256, 7, 7).transpose(0, 2, 3, 1)
        self._set_conv_krsc("roi_heads.box.feature_extractor.fc6", w6, None, sd["roi_heads.box.feature_extractor.fc6.bias"])
        w7 = sd["roi_heads.box.feature_extractor.fc7.weight"].reshape(1024, 1, 1, 1024)
        self._set_conv_krsc("roi_heads.box.feature_extractor.fc7", w7, None, sd["roi_heads.box.feature_extractor.fc7.bias"])
        wp = np.concatenate([sd["roi_heads.box.predictor.cls_score.weight"], sd["roi_heads.box.predictor.bbox_pred.weight"]], 0)
        bp = np.concatenate([sd["roi_heads.box.predictor.cls_score.bias"], sd["roi_heads.box.predictor.bbox_pred.bias"]])
        self._set_conv_krsc("roi_heads.box.predictor.cls_bbox", wp.reshape(405, 1, 1, 1024), None, bp)
        for i in range(1, 5):
            nm = "roi_heads.mask.feature_extractor.mask_fcn%d" % i
            self._set_conv_krsc(nm, to_krsc(sd[nm + ".weight"]), None, sd[nm + ".bias"])
        wd = sd["roi_heads.mask.predictor.conv5_mask.weight"]  # [Cin][Cout][2][2]
        for ab in range(4):
            wab = np.ascontiguousarray(wd[:, :, ab >> 1, ab & 1].T).reshape(256, 1, 1, 256)
            self._set_conv_krsc("roi_heads.mask.predictor.conv5_mask.%d" % ab, wab, None, sd["roi_heads.mask.predictor.conv5_mask.bias"])
        self._set_tensor("mask_logits.w", sd["roi_heads.mask.predictor.mask_fcn_logits.weight"].reshape(81, 256).astype(np.float32))
        self._set_tensor("mask_logits.b", sd["roi_heads.mask.predictor.mask_fcn_logits.bias"].astype(np.float32))
        # AnchorGenerator: the A base anchors per level go over; the engine lays the grid out for each batch's canvas (isegmi_op_grid_anchors)
        for l, (stride, size) in enumerate(zip(cfg.ANCHOR_STRIDE, cfg.ANCHOR_SIZES)):
            self._set_tensor("anchor_base.%d" % l, generate_anchors(stride, size, cfg.ASPECT_RATIOS))
            self.set_param("anchor_stride%d" % l, float(stride))

    def _load_bottlenecks(self, sd, src_prefix, dst_prefix, nblocks):
        for b in range(nblocks):
            src, dst = "%s.%d" % (src_prefix, b), "%s.%d" % (dst_prefix, b)
            for i in (1, 2, 3):
                self._set_conv_krsc("%s.conv%d" % (dst, i), to_krsc(sd["%s.conv%d.weight" % (src, i)]),
                                    *fold_frozen_batchnorm(sd, "%s.bn%d" % (src, i), self.cfg.FROZEN_BN_EPS))
            if b == 0:
                self._set_conv_krsc(dst + ".downsample.0", to_krsc(sd[src + ".downsample.0.weight"]),
                                    *fold_frozen_batchnorm(sd, src + ".downsample.1", self.cfg.FROZEN_BN_EPS))

    def _load_c4(self, sd):
        """R-50-C4 state dict (maskrcnn-benchmark names): backbone.body.{stem,layer1..3}, rpn.head.* (1024 ch, 15 anchors),
        roi_heads.box.feature_extractor.head.layer4.* (conv5 head, shared with the mask branch), roi_heads.box.predictor
        cls_score / bbox_pred on 2048, roi_heads.mask.predictor conv5_mask (2048 -> 256) / mask_fcn_logits."""
        cfg = self.cfg
        w = to_krsc(sd["backbone.body.stem.conv1.weight"])
        w = np.concatenate([w, np.zeros(w.shape[:3] + (1,), np.float32)], -1)
        self._set_conv_krsc("backbone.body.stem.conv1", w, *fold_frozen_batchnorm(sd, "backbone.body.stem.bn1", cfg.FROZEN_BN_EPS))
        for li, nb in enumerate((3, 4, 6), 1):
            self._load_bottlenecks(sd, "backbone.body.layer%d" % li, "backbone.body.layer%d" % li, nb)
        head = "roi_heads.box.feature_extractor.head.layer4"
        self._load_bottlenecks(sd, head, head, 3)
        self._set_conv_krsc("rpn.head.conv", to_krsc(sd["rpn.head.conv.weight"]), None, sd["rpn.head.conv.bias"])
        wcb = np.concatenate([to_krsc(sd["rpn.head.cls_logits.weight"]), to_krsc(sd["rpn.head.bbox_pred.weight"])], 0)
        bcb = np.concatenate([sd["rpn.head.cls_logits.bias"], sd["rpn.head.bbox_pred.bias"]])
        self._set_conv_krsc("rpn.head.cls_bbox", wcb, None, bcb)
        wp = np.concatenate([sd["roi_heads.box.predictor.cls_score.weight"], sd["roi_heads.box.predictor.bbox_pred.weight"]], 0)
        bp = np.concatenate([sd["roi_heads.box.predictor.cls_score.bias"], sd["roi_heads.box.predictor.bbox_pred.bias"]])
        self._set_conv_krsc("roi_heads.box.predictor.cls_bbox", wp.reshape(405, 1, 1, 2048), None, bp)
        wd = sd["roi_heads.mask.predictor.conv5_mask.weight"]  # [Cin 2048][Cout 256][2][2]
        for ab in range(4):
            wab = np.ascontiguousarray(wd[:, :, ab >> 1, ab & 1].T).reshape(256, 1, 1, wd.shape[0])
            self._set_conv_krsc("roi_heads.mask.predictor.conv5_mask.%d" % ab, wab, None, sd["roi_heads.mask.predictor.conv5_mask.bias"])
        self._set_tensor("mask_logits.w", sd["roi_heads.mask.predictor.mask_fcn_logits.weight"].reshape(81, 256).astype(np.float32))
        self._set_tensor("mask_logits.b", sd["roi_heads.mask.predictor.mask_fcn_logits.bias"].astype(np.float32))
        self._set_tensor("anchor_base.0", generate_anchors_multi(16, cfg.ANCHOR_SIZES, cfg.ASPECT_RATIOS))
        self.set_param("anchor_stride0", 16.0)

    # -- execution -----------------------------------------------------------------------------
    def _set_canvas(self, H, W):
        d = self.cfg.SIZE_DIVISIBILITY
        if H % d or W % d or H > self.H or W > self.W:
            raise ValueError("canvas %dx%d: must be a multiple of %d and fit inside the engine's %dx%d" % (H, W, d, self.H, self.W))
        self._canvas = (int(H), int(W))

    def upload(self, batch_nhwc3, image_hw, slot=0):
        """An already-normalised, zero-padded fp32 batch [n, H, W, 3] (prepare_images); its (H, W) is the canvas of the next forward."""
        x = np.ascontiguousarray(batch_nhwc3, np.float32)
        assert x.ndim == 4 and x.shape[3] == 3 and x.shape[0] <= self.max_batch, x.shape
        self._set_canvas(x.shape[1], x.shape[2])
        _ffi.check(_ffi.lib().isegmi_h2d(self.input_buffer(slot).ptr, x.ctypes.data_as(C.c_void_p), C.c_int64(x.nbytes)))
        self._hw = np.ascontiguousarray(image_hw, np.int32).reshape(-1, 2)
        assert self._hw.shape[0] == x.shape[0]
        return x.shape[0]

    def upload_u8(self, images_bgr_u8, slot=0, canvas=None):
        """to_image_list on the device (M1): a list of already-resized HxWx3 uint8 BGR images (what PIL's resize returns) -> mean-subtracted,
        zero-padded fp32 batch in the input buffer, bit-identical to prepare_images() on the host; a quarter of the PCIe bytes.  The batch's
        canvas is the padded maximum of its image sizes unless `canvas` = (H, W) forces a larger one."""
        assert 0 < len(images_bgr_u8) <= self.max_batch
        ims = [np.ascontiguousarray(im) for im in images_bgr_u8]
        if any(im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3 for im in ims):
            raise TypeError("upload_u8 takes HxWx3 uint8 images: a float batch goes through upload(batch, image_hw)")
        hw = np.array([im.shape[:2] for im in ims], np.int32)
        self._set_canvas(*(canvas or padded_canvas(hw, self.cfg.SIZE_DIVISIBILITY)))
        flat = np.concatenate([im.reshape(-1) for im in ims])
        st = self._u8_staging(slot, flat.nbytes)
        _ffi.check(_ffi.lib().isegmi_h2d(st.ptr, flat.ctypes.data_as(C.c_void_p), C.c_int64(flat.nbytes)))
        self._front_end(st, hw, slot)
        return len(ims)

    def _front_end(self, st, hw, slot):
        """build_transform's normalisation + to_image_list on the device: image i of the staging buffer -> plane i of the canvas batch."""
        H, W = self._canvas
        assert hw[:, 0].max() <= H and hw[:, 1].max() <= W, "image larger than the padded canvas"
        d_out = self.input_buffer(slot).ptr.value
        if hw.shape[0] > 1 and (hw == hw[0]).all():   # images of one size (a canvas-grouped batch, the bench batch): ONE launch for the batch
            h, w = int(hw[0, 0]), int(hw[0, 1])
            self._preprocess_u8(st.ptr.value, hw.shape[0], h, w, d_out, h, w, H, W, PIXEL_MEAN, (1.0, 1.0, 1.0), False)
            self._hw = hw
            return
        off = 0
        for i in range(hw.shape[0]):
            h, w = int(hw[i, 0]), int(hw[i, 1])
            self._preprocess_u8(st.ptr.value + off, 1, h, w, d_out + i * H * W * 3 * 4, h, w, H, W, PIXEL_MEAN, (1.0, 1.0, 1.0), False)
            off += h * w * 3
        self._hw = hw

    def upload_u8_async(self, pinned_u8, image_hw, slot=0, canvas=None):
        """upload_u8 from a uint8 _ffi.PinnedBuffer holding the images back to back (each h x w x 3), on the engine's copy stream."""
        hw = np.ascontiguousarray(image_hw, np.int32).reshape(-1, 2)
        nbytes = int((hw[:, 0].astype(np.int64) * hw[:, 1] * 3).sum())
        assert pinned_u8.nbytes >= nbytes and hw.shape[0] <= self.max_batch
        self._set_canvas(*(canvas or padded_canvas(hw, self.cfg.SIZE_DIVISIBILITY)))
        st = self._u8_staging(slot, nbytes)
        _ffi.check(_ffi.lib().isegmi_engine_upload_async(self._h, st.ptr, pinned_u8.ptr, C.c_int64(nbytes)))
        self._front_end(st, hw, slot)

    def forward_device(self, n, slot=0):
        H, W = self._canvas
        _ffi.check(_ffi.lib().isegmi_maskrcnn_forward_canvas(self._h, self.input_buffer(slot).ptr, self._hw.ctypes.data_as(C.c_void_p), n, H, W))

    def paste_device(self, out_h, out_w, orig_sizes_wh=None):
        """Masker paste into (out_h, out_w); orig_sizes_wh [N,2] are the sizes boxes are resized to (default: no resize)."""
        n = self._hw.shape[0]
        if orig_sizes_wh is None:
            ratios = np.ones((n, 2), np.float32)
        else:
            o = np.asarray(orig_sizes_wh, np.float64).reshape(n, 2)
            ratios = np.stack([o[:, 0] / self._hw[:, 1], o[:, 1] / self._hw[:, 0]], 1).astype(np.float32)
        _ffi.check(_ffi.lib().isegmi_maskrcnn_paste(self._h, ratios.ctypes.data_as(C.c_void_p), out_h, out_w))

    def sync(self):
        _ffi.check(_ffi.lib().isegmi_engine_sync(self._h))

    fetch = None  # bound below (shared with Yolact)
    timings = None

    def __call__(self, batch_nhwc3, image_hw=None):
        """-> list of BoxList (one per image, in network-input coordinates) with scores, labels, mask [n,1,28,28]
        (14x14 for the C4 predictor).  With image_hw = None the first argument is a list of already-resized uint8 BGR images and goes
        through the device front end (upload_u8: mean subtraction and padding on the GPU)."""
        n = self.upload_u8(batch_nhwc3) if image_hw is None else self.upload(batch_nhwc3, image_hw)
        self.forward_device(n)
        self.sync()
        cnt = self.fetch("det.count", n)
        box, score, label, m28 = (self.fetch(k, n) for k in ("det.box", "det.score", "det.label", self.mask_buf))
        out = []
        for i in range(n):
            c = int(cnt[i])
            bl = BoxList(box[i, :c], (int(self._hw[i, 1]), int(self._hw[i, 0])))
            bl.add_field("scores", score[i, :c]); bl.add_field("labels", label[i, :c].astype(np.int64))
            bl.add_field("mask", m28[i, :c, None])
            out.append(bl)
        return out

    def close(self):
        if self._h:
            _ffi.lib().isegmi_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from .yolact import Yolact as _Y  # noqa: E402

MaskRCNN.fetch = _Y.fetch
MaskRCNN.timings = _Y.timings
MaskRCNN.input_buffer = _Y.input_buffer
MaskRCNN.upload_async = _Y.upload_async
MaskRCNN.mark_step = _Y.mark_step
MaskRCNN._u8_staging = _Y._u8_staging
MaskRCNN._preprocess_u8 = _Y._preprocess_u8
MaskRCNN.step_times = _Y.step_times
for _n in ("wait_mark", "rle_device", "coco_record_bytes", "pack_coco_records", "download_async", "download_fence", "download_wait", "memory"):
    setattr(MaskRCNN, _n, getattr(_Y, _n))
