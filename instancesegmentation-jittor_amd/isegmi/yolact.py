"""Yolact R50-FPN inference, host side (mirrors Yolact.jittor's eval surface).

Reference surface (README.md:243-249): `python eval.py --trained_model=... --score_threshold=0.15
--top_k=15 --image=...` -> [UPSTREAM-RECALL, SURVEY 8b] `net(batch) -> [{'detection': {'box','mask',
'class','score','proto'}, 'net'}]` and `postprocess(dets, w, h, score_threshold) -> (classes, scores,
boxes, masks)`.  Everything numeric runs in libisegmi.so (HIP); this file only moves config,
weights and results across the C ABI.
"""
import ctypes as C
import math
from dataclasses import dataclass

import numpy as np

from . import _ffi
from .weights import dcn_blocks, fold_batchnorm, to_krsc

MEANS = (103.94, 116.78, 123.68)  # BGR
STD = (57.38, 57.12, 58.40)


@dataclass(frozen=True)
class YolactConfig:
    """yolact_resnet50_config constants."""
    max_size: int = 550
    num_classes: int = 81
    mask_dim: int = 32
    pred_scales: tuple = (24, 48, 96, 192, 384)
    pred_aspect_ratios: tuple = (1.0, 0.5, 2.0)
    nms_conf_thresh: float = 0.05
    nms_thresh: float = 0.5
    nms_top_k: int = 200
    max_num_detections: int = 100
    nms_second_threshold: int = 0  # SURVEY App. A.6 fork: Detect.fast_nms(second_threshold=True) -- a box must also have its own class score > nms_conf_thresh
    conv_split_k: int = 0          # opt-in latency numerics mode (see MaskRCNNConfig.CONV_SPLIT_K): fixed-tree split-K for the backbone's small-M / large-K convolutions
    depth: int = 50  # 50 = yolact_resnet50_config, 101 = yolact_base_config / yolact_im700_config
    # YOLACT++ (yolact_plus_*_config, the YOLACT++ rows of README.md:216-221): three scales per level x three aspect ratios,
    # rectangular anchors, DCNv2 3x3s in the backbone, fast mask re-scoring
    scales_per_level: int = 1          # 3: scale * 2**(j/3), j = 0..2
    use_square_anchors: bool = True
    dcn_layers: tuple = (0, 0, 0, 0)
    dcn_interval: int = 1
    use_maskiou: bool = False
    backbone: str = "resnet"           # "darknet53": yolact_darknet53_config (DarkNetBackbone([1, 2, 8, 8, 4]), LeakyReLU(0.1))

    @property
    def num_priors(self):
        return self.scales_per_level * len(self.pred_aspect_ratios)

    def level_scales(self, level):
        s = self.pred_scales[level]
        return tuple(s * 2 ** (j / 3.0) for j in range(self.scales_per_level))

    @staticmethod
    def darknet53():
        """yolact_darknet53_config: Darknet53-FPN at 550 (same heads / anchors as the ResNet configs; upstream preprocesses with
        darknet_transform: RGB, x / 255, no mean / std)."""
        return YolactConfig(backbone="darknet53")

    @staticmethod
    def plus_base():
        """yolact_plus_base_config: ResNet101-FPN with a DCNv2 every third block of layers 2-4 (resnet101_dcn_inter3_backbone)."""
        return YolactConfig(depth=101, scales_per_level=3, use_square_anchors=False, dcn_layers=(0, 4, 23, 3), dcn_interval=3,
                            use_maskiou=True)

    @staticmethod
    def plus_resnet50():
        """yolact_plus_resnet50_config: ResNet50-FPN with DCNv2 in every block of layers 2-4 (resnet50_dcnv2_backbone)."""
        return YolactConfig(depth=50, scales_per_level=3, use_square_anchors=False, dcn_layers=(0, 4, 6, 3), dcn_interval=1,
                            use_maskiou=True)

    @staticmethod
    def base():
        """yolact_base_config: ResNet101-FPN at 550."""
        return YolactConfig(depth=101)

    @staticmethod
    def im700():
        """yolact_im700_config: ResNet101-FPN at 700, pred_scales = int(s / 550 * 700)."""
        return YolactConfig(max_size=700, pred_scales=tuple(int(s / 550 * 700) for s in (24, 48, 96, 192, 384)), depth=101)


def make_priors(conv_h, conv_w, scales, max_size, ars, square=True):
    """PredictionModule.make_priors with use_pixel_scales, preapply_sqrt=False: per cell, scale-major / ratio-minor."""
    scales = tuple(scales) if isinstance(scales, (tuple, list)) else (scales,)
    out = np.empty((conv_h, conv_w, len(scales) * len(ars), 4), np.float64)
    xs = (np.arange(conv_w) + 0.5) / conv_w
    ys = (np.arange(conv_h) + 0.5) / conv_h
    out[..., 0] = xs[None, :, None]
    out[..., 1] = ys[:, None, None]
    a = 0
    for scale in scales:
        for ar in ars:
            r = math.sqrt(ar)
            w = scale * r / max_size
            out[..., a, 2] = w
            out[..., a, 3] = w if square else scale / r / max_size
            a += 1
    return out.reshape(-1, 4).astype(np.float32)


def darknet_base_transform(images_bgr_u8):
    """FastBaseTransform under darknet_transform (yolact_darknet53_config): x / 255, then BGR -> RGB; no mean / std.  NHWC3 fp32."""
    x = np.asarray(images_bgr_u8, np.float32) / np.float32(255.0)
    return np.ascontiguousarray(x[..., ::-1])


def fast_base_transform(images_bgr_u8):
    """FastBaseTransform for already-550x550 BGR uint8 images (Y1): (x-mean)/std then BGR->RGB. NHWC3 fp32."""
    x = np.asarray(images_bgr_u8, np.float32)
    x = (x - np.asarray(MEANS, np.float32)) / np.asarray(STD, np.float32)
    return np.ascontiguousarray(x[..., ::-1])


def _level_sizes(size):
    s = (size + 6 - 7) // 2 + 1      # conv1 7x7/2 p3
    s = (s + 2 - 3) // 2 + 1         # maxpool 3/2/1
    c3 = (s + 2 - 3) // 2 + 1        # layer2 stride 2 (3x3 p1)
    c4 = (c3 + 2 - 3) // 2 + 1
    c5 = (c4 + 2 - 3) // 2 + 1
    p6 = (c5 + 2 - 3) // 2 + 1
    p7 = (p6 + 2 - 3) // 2 + 1
    return [c3, c4, c5, p6, p7]


class Yolact:
    """`net = Yolact(state_dict); preds = net(batch)`; `postprocess(preds, w, h, score_threshold)`."""

    KIND = 1

    def __init__(self, state_dict, cfg=YolactConfig(), max_batch=8, device=0, input_size=None, fuse_heads=True, fp16=False):
        """fp16=True: optional fp16-storage / f16-MFMA mode (backbone, FPN, protonet and head convolutions; the fused head
        outputs, the prototypes and Detect / mask assembly stay fp32).  Not bit-exact: tolerance parity vs YolactRef(fp16=True)."""
        assert not fp16 or fuse_heads, "fp16 Yolact uses the fused prediction head"
        self.fp16 = bool(fp16)
        self.cfg = cfg
        self.size = int(input_size or cfg.max_size)
        self.max_batch = max_batch
        self.fuse_heads = bool(fuse_heads)
        L = _ffi.lib()
        _ffi.set_device(device)
        self._h = C.c_void_p()
        _ffi.check(L.isegmi_engine_create(self.KIND, max_batch, self.size, self.size, C.byref(self._h)))
        self.set_param("resnet_depth", float(cfg.depth))
        self.set_param("num_priors", float(cfg.num_priors))
        self.set_param("darknet", 1.0 if cfg.backbone == "darknet53" else 0.0)
        assert not (fp16 and cfg.backbone == "darknet53"), "the Darknet53 backbone runs in fp32 only"
        assert not (fp16 and any(cfg.dcn_layers)), "the DCNv2 backbones run in fp32 only"
        if self.fp16:  # must precede weight loading (weights are packed as fp16)
            self.set_param("fp16", 1.0)
        self._load(state_dict)
        for k in ("nms_conf_thresh", "nms_thresh", "nms_top_k", "max_num_detections", "nms_second_threshold", "conv_split_k"):
            self.set_param(k, float(getattr(cfg, k)))
        self._d_in = _ffi.DeviceBuffer((max_batch, self.size, self.size, 3))
        self._forward_id = 0
        self._pp_key = None

    # -- weights -------------------------------------------------------------------------------
    def _set_conv(self, name, w_oihw, scale=None, shift=None, pad_cin_to=None):
        w = to_krsc(w_oihw)
        if pad_cin_to and w.shape[3] < pad_cin_to:
            w = np.concatenate([w, np.zeros(w.shape[:3] + (pad_cin_to - w.shape[3],), np.float32)], -1)
        cout, r, s, cin = w.shape
        fp = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).ctypes.data_as(C.c_void_p)
        sc = None if scale is None else np.ascontiguousarray(scale, np.float32)
        sh = None if shift is None else np.ascontiguousarray(shift, np.float32)
        _ffi.check(_ffi.lib().isegmi_engine_set_conv(self._h, name.encode(), cout, r, s, cin, fp(w), fp(sc), fp(sh)))

    def _load(self, sd):
        dcn = dcn_blocks(self.cfg.depth, self.cfg.dcn_layers, self.cfg.dcn_interval)
        if self.cfg.backbone == "darknet53":
            sc, sh = fold_batchnorm(sd, "backbone._preconv.1")
            self._set_conv("backbone._preconv.0", sd["backbone._preconv.0.weight"], sc, sh, pad_cin_to=32)
            for li, nb in enumerate((1, 2, 8, 8, 4)):
                nm = "backbone.layers.%d" % li
                sc, sh = fold_batchnorm(sd, nm + ".0.1")
                self._set_conv(nm + ".0.0", sd[nm + ".0.0.weight"], sc, sh)
                for b in range(1, nb + 1):
                    for i in (1, 2):
                        sc, sh = fold_batchnorm(sd, "%s.%d.bn%d" % (nm, b, i))
                        self._set_conv("%s.%d.conv%d" % (nm, b, i), sd["%s.%d.conv%d.weight" % (nm, b, i)], sc, sh)
        else:
            self._load_resnet(sd, dcn)
        self._load_heads(sd)

    def _load_resnet(self, sd, dcn):
        sc, sh = fold_batchnorm(sd, "backbone.bn1")
        self._set_conv("backbone.conv1", sd["backbone.conv1.weight"], sc, sh, pad_cin_to=4)
        for li, nb in enumerate((3, 4, 23 if self.cfg.depth == 101 else 6, 3)):
            for b in range(nb):
                nm = "backbone.layers.%d.%d" % (li, b)
                for i in (1, 2, 3):
                    sc, sh = fold_batchnorm(sd, "%s.bn%d" % (nm, i))
                    w = sd["%s.conv%d.weight" % (nm, i)]
                    if i == 2 and (li, b) in dcn:
                        # DCNv2: the engine samples the 9 taps into columns (isegmi_op_deform_im2col) and runs the 3x3 as a 1x1 over
                        # 9*Cin channels in KRSC order -- the same k-ordered chain; the DCN bias folds into the BN shift
                        k = to_krsc(w)
                        w = k.reshape(k.shape[0], 1, 1, -1).transpose(0, 3, 1, 2)
                        sh = (sh + sd[nm + ".conv2.bias"].astype(np.float32) * sc).astype(np.float32)
                        self._set_conv(nm + ".conv2.conv_offset_mask", sd[nm + ".conv2.conv_offset_mask.weight"], None,
                                       sd[nm + ".conv2.conv_offset_mask.bias"])
                    self._set_conv("%s.conv%d" % (nm, i), w, sc, sh)
                if b == 0:
                    sc, sh = fold_batchnorm(sd, nm + ".downsample.1")
                    self._set_conv(nm + ".downsample.0", sd[nm + ".downsample.0.weight"], sc, sh)

    def _load_heads(self, sd):
        biased = ["fpn.lat_layers.%d" % i for i in range(3)] + ["fpn.pred_layers.%d" % i for i in range(3)] + \
                 ["fpn.downsample_layers.%d" % i for i in range(2)] + ["proto_net.%d" % i for i in (0, 2, 4, 8, 10)] + \
                 ["prediction_layers.0." + n for n in ("upfeature.0", "bbox_layer", "conf_layer", "mask_layer")]
        for nm in biased:
            self._set_conv(nm, sd[nm + ".weight"], None, sd[nm + ".bias"])
        # fused prediction head: [A*4 loc | A*81 conf | A*32 mask] as one 351-wide conv (tanh moves to the gather)
        if self.fuse_heads:
            names = ["prediction_layers.0." + n for n in ("bbox_layer", "conf_layer", "mask_layer")]
            self._set_conv("prediction_layers.0.head_cat", np.concatenate([sd[n + ".weight"] for n in names], 0), None,
                           np.concatenate([sd[n + ".bias"] for n in names]))
        self.has_maskiou = bool(self.cfg.use_maskiou and "maskiou_net.0.weight" in sd)  # a checkpoint without the net: box scores only
        if self.has_maskiou:
            # FastMaskIoUNet: layer 0 (one input channel) goes over as plain tensors for its dedicated kernel; layers 2..8 and the
            # 1x1 run on the MFMA conv kernels with channels padded to 32 by ZERO weights / biases (exact: the extra terms are +-0)
            w0 = to_krsc(sd["maskiou_net.0.weight"]).reshape(8, 9)
            for nm, arr in (("maskiou.w0", w0), ("maskiou.b0", sd["maskiou_net.0.bias"])):
                self._set_tensor(nm, np.ascontiguousarray(arr, np.float32))
            for i in (2, 4, 6, 8, 10):
                w = np.asarray(sd["maskiou_net.%d.weight" % i], np.float32); bias = np.asarray(sd["maskiou_net.%d.bias" % i], np.float32)
                cout, cin = w.shape[:2]
                cin_p = max(32, cin); cout_p = max(32, cout) if i != 10 else cout
                wp = np.zeros((cout_p, cin_p) + w.shape[2:], np.float32); wp[:cout, :cin] = w
                bp = np.zeros(cout_p, np.float32); bp[:cout] = bias
                self._set_conv("maskiou_net.%d" % i, wp, None, bp)
        pri = [make_priors(s, s, self.cfg.level_scales(l), self.cfg.max_size, self.cfg.pred_aspect_ratios, self.cfg.use_square_anchors)
               for l, s in enumerate(_level_sizes(self.size))]
        self.priors = np.concatenate(pri, 0)
        self._set_tensor("priors", self.priors)

    def _set_tensor(self, name, a):
        a = np.ascontiguousarray(a)
        _ffi.check(_ffi.lib().isegmi_engine_set_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), C.c_int64(a.nbytes)))

    def set_param(self, name, value):
        _ffi.check(_ffi.lib().isegmi_engine_set_param(self._h, name.encode(), C.c_float(value)))

    # -- execution -----------------------------------------------------------------------------
    def upload(self, batch_nhwc3):
        x = np.ascontiguousarray(batch_nhwc3, np.float32)
        assert x.ndim == 4 and x.shape[1:] == (self.size, self.size, 3) and x.shape[0] <= self.max_batch, x.shape
        _ffi.check(_ffi.lib().isegmi_h2d(self._d_in.ptr, x.ctypes.data_as(C.c_void_p), C.c_int64(x.nbytes)))
        return x.shape[0]


    def input_buffer(self, slot=0):
        """Device input buffer `slot` (0 or 1; the second one is allocated on first use, for double-buffered uploads)."""
        if slot == 0:
            return self._d_in
        if getattr(self, "_d_in2", None) is None:
            self._d_in2 = _ffi.DeviceBuffer(self._d_in.shape)
        return self._d_in2

    def upload_async(self, pinned, slot=0):
        """Asynchronous H2D of a whole batch from a _ffi.PinnedBuffer into input buffer `slot` on the engine's copy stream; the
        next forward is ordered behind it and it is ordered behind the previous forward's read of its input."""
        d = self.input_buffer(slot)
        assert pinned.nbytes <= d.nbytes
        _ffi.check(_ffi.lib().isegmi_engine_upload_async(self._h, d.ptr, pinned.ptr, C.c_int64(pinned.nbytes)))

    # -- device front end: bytes go over PCIe, the transform runs on the engine's stream ------------------------------------
    def _u8_staging(self, slot, nbytes):
        st = getattr(self, "_u8", None)
        if st is None:
            st = self._u8 = [None, None]
        if st[slot] is None or st[slot].nbytes < nbytes:
            if st[slot] is not None:
                self.sync()  # a queued transform may still read the old buffer
                st[slot].free()
            st[slot] = _ffi.DeviceBuffer((int(nbytes),), np.uint8)
        return st[slot]

    def _preprocess_u8(self, d_u8_ptr, n, hin, win, d_out_ptr, hout, wout, hpad, wpad, mean, std, swap_rb):
        m = (C.c_float * 3)(*[float(v) for v in mean]); sd = (C.c_float * 3)(*[float(v) for v in std])
        _ffi.check(_ffi.lib().isegmi_engine_preprocess_u8(self._h, C.c_void_p(d_u8_ptr), n, hin, win, C.c_void_p(d_out_ptr), hout, wout, hpad, wpad,
                                                          C.c_int64(hpad * wpad * 3), m, sd, 1 if swap_rb else 0))

    def _u8_norm(self):
        darknet = self.cfg.backbone == "darknet53"
        return ((0.0, 0.0, 0.0), (255.0, 255.0, 255.0)) if darknet else (MEANS, STD)

    def upload_u8(self, images_bgr_u8, slot=0):
        """FastBaseTransform on the device (Y1): uint8 BGR images -- an [N, H, W, 3] array or a list of HxWx3 arrays of DIFFERENT sizes --
        -> bilinear resize to the network size, the backbone's normalisation, RGB: bit-identical to isegmi.transforms.yolact_transform
        on the host, a quarter of the PCIe bytes (and none of the host's resize arithmetic).  The previous forward on this input slot
        must have completed, as for upload()."""
        ims = self._as_u8_list(images_bgr_u8)
        flat = ims[0].reshape(-1) if len(ims) == 1 else np.concatenate([im.reshape(-1) for im in ims])
        st = self._u8_staging(slot, flat.nbytes)
        _ffi.check(_ffi.lib().isegmi_h2d(st.ptr, flat.ctypes.data_as(C.c_void_p), C.c_int64(flat.nbytes)))
        self._front_end(st, [im.shape[:2] for im in ims], slot)
        return len(ims)

    def _as_u8_list(self, images):
        if isinstance(images, np.ndarray) and images.ndim == 4:
            images = [images[i] for i in range(images.shape[0])] if images.shape[0] != 1 else [images[0]]
        ims = [np.ascontiguousarray(im) for im in images]
        if any(im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3 for im in ims):
            raise TypeError("upload_u8 takes HxWx3 uint8 images: a float batch is the already-transformed input of upload()")
        assert 0 < len(ims) <= self.max_batch, len(ims)
        return ims

    def _front_end(self, st, sizes_hw, slot):
        """image i of the staging buffer (back to back, each h x w x 3) -> plane i of the network input; equal sizes go as one launch"""
        mean, std = self._u8_norm()
        d_out = self.input_buffer(slot).ptr.value
        plane = self.size * self.size * 3 * 4
        sizes = [(int(h), int(w)) for h, w in sizes_hw]
        if len(set(sizes)) == 1:
            h, w = sizes[0]
            self._preprocess_u8(st.ptr.value, len(sizes), h, w, d_out, self.size, self.size, self.size, self.size, mean, std, True)
            return
        off = 0
        for i, (h, w) in enumerate(sizes):
            self._preprocess_u8(st.ptr.value + off, 1, h, w, d_out + i * plane, self.size, self.size, self.size, self.size, mean, std, True)
            off += h * w * 3

    def upload_u8_async(self, pinned_u8, n, h=None, w=None, slot=0, sizes_hw=None):
        """upload_u8 from a uint8 _ffi.PinnedBuffer holding the images back to back, on the engine's copy stream (see upload_async for
        the ordering): n images of one size (h, w), or sizes_hw = [(h, w), ...] per image."""
        sizes = [(int(h), int(w))] * n if sizes_hw is None else [(int(a), int(b)) for a, b in sizes_hw]
        nbytes = sum(a * b * 3 for a, b in sizes)
        assert pinned_u8.nbytes >= nbytes and len(sizes) <= self.max_batch
        st = self._u8_staging(slot, nbytes)
        _ffi.check(_ffi.lib().isegmi_engine_upload_async(self._h, st.ptr, pinned_u8.ptr, C.c_int64(nbytes)))
        self._front_end(st, sizes, slot)

    def mark_step(self):
        _ffi.check(_ffi.lib().isegmi_engine_mark_step(self._h))

    def wait_mark(self, back=0):
        """host wait for the completion mark `back` marks before the newest one: bounds the steps a producer loop keeps in flight"""
        _ffi.check(_ffi.lib().isegmi_engine_wait_mark(self._h, int(back)))

    def step_times(self, cap=65536):
        """Intervals (ms) between consecutive mark_step() completion marks; clears the marks."""
        ms = (C.c_float * cap)(); cnt = C.c_int()
        _ffi.check(_ffi.lib().isegmi_engine_step_times(self._h, ms, cap, C.byref(cnt)))
        return [float(ms[i]) for i in range(cnt.value)]

    def forward_device(self, n, slot=0):
        """Launch forward on the batch already resident in the engine's input buffer (asynchronous)."""
        self._pp_key = None  # masks / integer boxes cached by postprocess() belong to the previous forward
        self._forward_id += 1
        _ffi.check(_ffi.lib().isegmi_yolact_forward(self._h, self.input_buffer(slot).ptr, n))

    def postprocess_device(self, h, w):
        _ffi.check(_ffi.lib().isegmi_yolact_postprocess(self._h, h, w))

    def postprocess_device_sizes(self, image_hw):
        """postprocess with image n assembled at ITS (h, w) = image_hw[n] inside a common plane of the batch's maximum size."""
        hw = np.ascontiguousarray(image_hw, np.int32).reshape(-1, 2)
        _ffi.check(_ffi.lib().isegmi_yolact_postprocess_sizes(self._h, hw.ctypes.data_as(C.c_void_p), hw.shape[0]))

    def sync(self):
        _ffi.check(_ffi.lib().isegmi_engine_sync(self._h))

    # -- device-side COCO output (shared with MaskRCNN): RLE of the masks, one fixed-size record block per batch ---------------------
    def rle_device(self, image_hw=None):
        """pycocotools RLE (counts + compressed strings) of det.masks of the last postprocess / paste, on the results stream; image_hw
        [n, 2] = every image's own (h, w) inside the mask planes (None: the whole plane)."""
        hw = None if image_hw is None else np.ascontiguousarray(image_hw, np.int32).reshape(-1, 2)
        _ffi.check(_ffi.lib().isegmi_engine_rle(self._h, None if hw is None else hw.ctypes.data_as(C.c_void_p)))

    def coco_record_bytes(self, n):
        """(total bytes, offset of the strings) of the record block of a batch of n images (isegmi_engine_pack_coco_records)."""
        nb, co = C.c_int64(), C.c_int64()
        _ffi.check(_ffi.lib().isegmi_engine_coco_record_bytes(self._h, n, C.byref(nb), C.byref(co)))
        return nb.value, co.value

    def pack_coco_records(self, dev_buffer, n_block):
        nb = C.c_int64()
        _ffi.check(_ffi.lib().isegmi_engine_pack_coco_records(self._h, dev_buffer.ptr, C.c_int64(dev_buffer.nbytes), int(n_block), C.byref(nb)))
        return nb.value

    def download_async(self, slot, pinned, dev_buffer, nbytes):
        _ffi.check(_ffi.lib().isegmi_engine_download_async(self._h, slot, pinned.ptr, dev_buffer.ptr, C.c_int64(nbytes)))

    def download_fence(self, slot):
        _ffi.check(_ffi.lib().isegmi_engine_download_fence(self._h, slot))

    def download_wait(self, slot):
        _ffi.check(_ffi.lib().isegmi_engine_download_wait(self._h, slot))

    def memory(self):
        """(weight_bytes, buffer_bytes) of device memory held by the engine (+ the input slots owned by this wrapper)."""
        wb, bb = C.c_int64(), C.c_int64()
        _ffi.check(_ffi.lib().isegmi_engine_memory(self._h, C.byref(wb), C.byref(bb)))
        extra = self._d_in.nbytes + (self._d_in2.nbytes if getattr(self, "_d_in2", None) is not None else 0)
        return wb.value, bb.value + extra

    def fetch(self, name, rows=None):
        """D2H copy of a named engine buffer (optionally only its first `rows` leading rows)."""
        p = C.c_void_p(); nb = C.c_int64(); dt = C.c_int32(); nd = C.c_int32(); shp = (C.c_int64 * 4)()
        _ffi.check(_ffi.lib().isegmi_engine_buffer_info(self._h, name.encode(), C.byref(p), C.byref(nb), C.byref(dt), shp, C.byref(nd)))
        dtype = [np.float32, np.int32, np.uint8, np.int64, np.float16][dt.value]
        shape = tuple(int(shp[i]) for i in range(nd.value))
        if rows is not None:
            shape = (rows,) + shape[1:]
        out = np.empty(shape, dtype)
        if out.nbytes:
            _ffi.check(_ffi.lib().isegmi_d2h(out.ctypes.data_as(C.c_void_p), p, C.c_int64(out.nbytes)))
        return out

    def timings(self):
        names = C.create_string_buffer(4096); ms = (C.c_float * 64)(); cnt = C.c_int()
        _ffi.check(_ffi.lib().isegmi_engine_get_timings(self._h, names, 4096, ms, 64, C.byref(cnt)))
        ns = names.value.decode().split(";") if cnt.value else []
        return [(ns[i], float(ms[i])) for i in range(cnt.value)]

    def __call__(self, batch_nhwc3):
        """Upstream-shaped result: list (one per image) of {'detection': {...}|None, 'net': self}.  A uint8 [N, H, W, 3] batch is taken as
        raw BGR images of any one size and goes through the device front end (upload_u8: FastBaseTransform on the GPU); a float batch is
        the already-transformed network input."""
        raw = isinstance(batch_nhwc3, (list, tuple)) or np.asarray(batch_nhwc3).dtype == np.uint8
        n = self.upload_u8(batch_nhwc3) if raw else self.upload(batch_nhwc3)
        self.forward_device(n)
        self.sync()
        cnt = self.fetch("det.count", n)
        box, score, cls, coeff, prior = (self.fetch(k, n) for k in ("det.box", "det.score", "det.class", "det.coeff", "det.prior"))
        proto = self.fetch("proto", n)
        out = []
        for i in range(n):
            c = int(cnt[i])
            det = None if c == 0 else dict(box=box[i, :c], mask=coeff[i, :c], score=score[i, :c], proto=proto[i],
                                           prior=prior[i, :c], **{"class": cls[i, :c]})
            out.append({"detection": det, "net": self, "_index": i, "_forward": self._forward_id})
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


def postprocess(det_output, w, h, batch_idx=0, score_threshold=0.0):
    """Yolact layers/output_utils.postprocess (Y7): (classes, scores, boxes[int64 xyxy], masks[n,h,w] uint8).

    Masks are produced on the GPU for the whole last batch at (h, w); `score_threshold` filters the
    (score-sorted) detections afterwards, which is equivalent because the filter precedes all
    arithmetic upstream and every detection's mask is independent."""
    dets = det_output[batch_idx]
    net = dets["net"]
    if dets["detection"] is None:
        z = np.zeros((0,), np.int32)
        return z, np.zeros((0,), np.float32), np.zeros((0, 4), np.int64), np.zeros((0, h, w), np.uint8)
    i = dets["_index"]
    if dets.get("_forward", net._forward_id) != net._forward_id:
        raise RuntimeError("postprocess(): these predictions come from an earlier forward of this net; the device buffers "
                           "(prototypes, coefficients) now hold a later batch")
    key = (h, w, net._forward_id)
    if getattr(net, "_pp_key", None) != key:
        net.postprocess_device(h, w)
        net.sync()
        net._pp_key = key
        net._pp_masks = net.fetch("det.masks")
        net._pp_boxes = net.fetch("det.box_int")
        net._pp_mask_scores = net.fetch("det.mask_score") if net.has_maskiou else None
    d = dets["detection"]
    keep = d["score"] > np.float32(score_threshold)
    c = len(d["score"])
    scores = d["score"][keep]
    if net.has_maskiou:  # upstream (rescore_mask, not rescore_bbox): scores = [box scores, box scores * mask IoU]
        scores = [scores, net._pp_mask_scores[i, :c][keep]]
    return d["class"][keep], scores, net._pp_boxes[i, :c][keep], net._pp_masks[i, :c][keep]


def evaluate(net, images, image_ids=None, batch_size=None, score_threshold=0.0, top_k=None, rank=0, world=1, sizes=None, stats=None, force_gather=False):
    """eval.py's image evaluation as a data-set loop (README.md:243-249: --images / --output_coco_json): images -> COCO-format result list
    (Detections.add_bbox / add_mask records) -- the path the benchmark measures, end to end.  Batches of `batch_size` raw uint8 BGR
    images (any sizes) go up through pinned memory; FastBaseTransform, the forward, Detect, postprocess at every image's OWN size,
    pycocotools RLE and the packing of one fixed-size record block run on the device; the block comes back asynchronously (one rank) or
    is all-gathered over RCCL (`world` ranks, batches round-robin) while the next batch computes.

    images: sequence of HxWx3 uint8 BGR arrays, or a callable i -> array together with sizes = [(h, w), ...]."""
    import time
    from .coco import results_from_records
    from .pipeline import record_capacity, run_record_loop
    get = images if callable(images) else images.__getitem__
    if sizes is None:
        if callable(images):
            raise ValueError("evaluate(): a callable image source needs sizes=[(h, w), ...]")
        sizes = [im.shape[:2] for im in images]
    n_img = len(sizes)
    ids = list(image_ids) if image_ids is not None else list(range(n_img))
    bs = int(batch_size or net.max_batch)
    assert bs <= net.max_batch
    batches = [list(range(j, min(j + bs, n_img))) for j in range(0, n_img, bs)]
    K = record_capacity(net)
    pin_bytes = max(sum(sizes[i][0] * sizes[i][1] * 3 for i in b) for b in batches) if batches else 1
    pin = [_ffi.PinnedBuffer((pin_bytes,), np.uint8) for _ in range(2)]
    for slot in (0, 1):
        net._u8_staging(slot, pin_bytes)   # both device staging buffers at their final size: no sync + re-allocation when a later batch is larger
    per_image = [None] * n_img

    def consume(step, recs):
        for r, rec in enumerate(recs):
            j = step * world + r
            if j >= len(batches):
                continue
            b = batches[j]
            res = results_from_records(rec, [ids[i] for i in b] + [None] * (bs - len(b)), [sizes[i] for i in b] + [(1, 1)] * (bs - len(b)), 1,
                                       K, score_threshold, top_k)
            by_id = {}
            for d in res:
                by_id.setdefault(d["image_id"], []).append(d)
            for i in b:
                per_image[i] = by_id.get(ids[i], [])

    def enqueue(step, slot):
        j = step * world + rank
        if j >= len(batches):
            return False
        b = batches[j]
        off = 0
        for i in b:
            im = np.ascontiguousarray(get(i), np.uint8)
            pin[slot].array[off:off + im.size] = im.reshape(-1)
            off += im.size
        hw = [sizes[i] for i in b]
        net.upload_u8_async(pin[slot], len(b), slot=slot, sizes_hw=hw)
        net.forward_device(len(b), slot)
        net.postprocess_device_sizes(hw)
        net.rle_device(hw)
        return True

    t0 = time.perf_counter()
    nsteps = -(-len(batches) // world)
    try:
        run_record_loop(net, bs, nsteps, enqueue, consume, rank, world, force_gather)   # closes its pipeline / gather on every way out
    finally:
        for p in pin:
            p.free()
    if stats is not None:
        stats.update(steps=nsteps, images=n_img, seconds=time.perf_counter() - t0, batches=len(batches), batch_size=bs, world=world)
    return [d for r in per_image if r for d in r]
