"""CPU oracle of the Yolact R50-FPN forward + Detect + postprocess (TEST INFRASTRUCTURE ONLY).

numpy graph over the C oracle ops (oracle/ora_ops.c).  Follows SURVEY.md Appendix A.9 / A.1 /
A.6 ([UPSTREAM-RECALL] of dbolya/yolact, the lineage the reference names at README.md:355; the
reference's own Yolact.jittor sources are absent -> PARITY UNPINNED).  Takes the SAME upstream-
named state dict (OIHW weights, BN running stats) as the product and folds BN itself.
"""
import math

import numpy as np

from . import ora


def _krsc(w_oihw):
    return np.ascontiguousarray(np.transpose(np.asarray(w_oihw, np.float32), (0, 2, 3, 1)))


def _fold_bn(sd, prefix, eps=np.float32(1e-5)):
    """BatchNorm2d eval (A.1): y = (x-mean)/sqrt(var+eps)*w + b  ==  x*scale + shift."""
    w = sd[prefix + ".weight"].astype(np.float32); b = sd[prefix + ".bias"].astype(np.float32)
    m = sd[prefix + ".running_mean"].astype(np.float32); v = sd[prefix + ".running_var"].astype(np.float32)
    scale = (w / np.sqrt(v + eps)).astype(np.float32)
    shift = (b - m * scale).astype(np.float32)
    return scale, shift


def make_priors(conv_h, conv_w, scale, max_size, ars=(1.0, 0.5, 2.0), square=True):
    """A.9 make_priors with use_pixel_scales, preapply_sqrt=False; scale: one number (use_square_anchors configs) or the
    level's list of scales (YOLACT++: three per level, rectangular anchors), scale-major / ratio-minor per cell."""
    scales = tuple(scale) if isinstance(scale, (tuple, list)) else (scale,)
    out = []
    for j in range(conv_h):
        for i in range(conv_w):
            x = (i + 0.5) / conv_w
            y = (j + 0.5) / conv_h
            for sc in scales:
                for ar in ars:
                    ar = math.sqrt(ar)
                    w = sc * ar / max_size
                    h = w if square else sc / ar / max_size
                    out += [x, y, w, h]
    return np.asarray(out, np.float64).astype(np.float32).reshape(-1, 4)


class YolactRef:
    def __init__(self, sd, max_size=550, scales=(24, 48, 96, 192, 384), depth=50, fp16=False, scales_per_level=1, square=True,
                 second_threshold=0, conv_split_k=0):
        # YOLACT++: scales_per_level=3, square=False; DCNv2 blocks and the mask-IoU net are recognised by their state-dict
        # entries (<block>.conv2.conv_offset_mask.weight, maskiou_net.0.weight)
        self.scales_per_level = scales_per_level
        self.square = square
        self.second_threshold = int(second_threshold)   # App. A.6 fork: Detect.fast_nms(second_threshold=True)
        # the product's opt-in `conv_split_k` numerics mode: conv2 / conv3 of every bottleneck and conv1 of a stage's later blocks are evaluated as FOUR
        # k-ordered partial chains added left to right (ora.conv2d(ksplit=4)) where the shape rule takes the layer (ora.conv_split_qualifies)
        self.conv_split_k = int(conv_split_k)
        # fp16=True emulates the product's optional fp16-storage mode: image, conv weights and every stored activation are rounded
        # to fp16 (the fused head outputs and the prototypes stay fp32), arithmetic stays the fp32 ordered chain.
        self.fp16 = fp16
        self.sd = sd
        self.max_size = max_size
        self.scales = tuple(scales)
        self.depth = depth
        self.feats = {}

    def _h(self, x):
        return x.astype(np.float16).astype(np.float32) if self.fp16 else x

    def _conv_bn(self, x, name, bn, stride, pad, act, residual=None, may_split=False):
        sc, sh = _fold_bn(self.sd, bn)
        w = self._h(_krsc(self.sd[name + ".weight"]))
        ks = 1
        if may_split and self.conv_split_k and not self.fp16:
            ho, wo = (x.shape[1] + 2 * pad - w.shape[1]) // stride + 1, (x.shape[2] + 2 * pad - w.shape[2]) // stride + 1
            ks = 4 if ora.conv_split_qualifies(x.shape[0] * ho * wo, w.shape[0], w.shape[1], w.shape[2], w.shape[3]) else 1
        return self._h(ora.conv2d(x, w, stride, pad, sc, sh, residual, act, ksplit=ks))

    def _conv_b(self, x, name, stride, pad, act, keep_f32=False, **kw):
        y = ora.conv2d(x, self._h(_krsc(self.sd[name + ".weight"])), stride, pad, None, self.sd[name + ".bias"], None, act, **kw)
        return y if keep_f32 else self._h(y)

    def _darknet(self, x):
        """DarkNetBackbone([1, 2, 8, 8, 4]) of yolact_darknet53_config [UPSTREAM-RECALL]: every conv is Conv(bias=False) + BN +
        LeakyReLU(0.1) (conv2d act 3); a block is 1x1 (C -> C/2), 3x3 (C/2 -> C) and the shortcut added AFTER the activation
        (act 4); layers 2, 3, 4 (256 / 512 / 1024 channels at strides 8 / 16 / 32) feed the FPN."""
        x = self._conv_bn(x, "backbone._preconv.0", "backbone._preconv.1", 1, 1, 3)
        outs = []
        for li, nb in enumerate((1, 2, 8, 8, 4)):
            nm = "backbone.layers.%d" % li
            x = self._conv_bn(x, nm + ".0.0", nm + ".0.1", 2, 1, 3)
            for b in range(1, nb + 1):
                t = self._conv_bn(x, "%s.%d.conv1" % (nm, b), "%s.%d.bn1" % (nm, b), 1, 0, 3)
                x = self._conv_bn(t, "%s.%d.conv2" % (nm, b), "%s.%d.bn2" % (nm, b), 1, 1, 4, residual=x)
            outs.append(x)
        return outs[2], outs[3], outs[4]

    def forward(self, images_nhwc3):
        x = np.asarray(images_nhwc3, np.float32)
        N = x.shape[0]
        if "backbone._preconv.0.weight" in self.sd:
            C3, C4, C5 = self._darknet(x)
            return self._heads(N, C3, C4, C5)
        x4 = np.concatenate([x, np.zeros(x.shape[:3] + (1,), np.float32)], -1)
        w1 = _krsc(self.sd["backbone.conv1.weight"])
        w1 = np.concatenate([w1, np.zeros(w1.shape[:3] + (1,), np.float32)], -1)
        sc, sh = _fold_bn(self.sd, "backbone.bn1")
        x = self._h(ora.conv2d(self._h(x4), self._h(w1), 2, 3, sc, sh, None, 1))
        x = ora.maxpool(x, 3, 2, 1)
        outs = []
        for li, nb in enumerate((3, 4, 23 if self.depth == 101 else 6, 3)):
            for b in range(nb):
                nm = "backbone.layers.%d.%d" % (li, b)
                st = 2 if (b == 0 and li > 0) else 1
                idt = x
                if b == 0:
                    idt = self._conv_bn(x, nm + ".downsample.0", nm + ".downsample.1", st, 0, 0)
                t = self._conv_bn(x, nm + ".conv1", nm + ".bn1", 1, 0, 1, may_split=b > 0)
                if nm + ".conv2.conv_offset_mask.weight" in self.sd:
                    # DCNv2 (modulated deformable 3x3): offsets / mask logits from a plain 3x3 on the same input, the nine taps
                    # sampled bilinearly (ora_deform_im2col), then the weights applied in (r, s, cin) order; bias, then BN + ReLU
                    om = self._conv_b(t, nm + ".conv2.conv_offset_mask", st, 1, 0, keep_f32=True)
                    col = ora.deform_im2col(t, om, 3, 3, st, 1, 1)
                    w = _krsc(self.sd[nm + ".conv2.weight"])
                    sc, sh = _fold_bn(self.sd, nm + ".bn2")
                    sh = (sh + self.sd[nm + ".conv2.bias"].astype(np.float32) * sc).astype(np.float32)
                    t = ora.conv2d(col, w.reshape(w.shape[0], 1, 1, -1), 1, 0, sc, sh, None, 1)
                else:
                    t = self._conv_bn(t, nm + ".conv2", nm + ".bn2", st, 1, 1, may_split=True)
                x = self._conv_bn(t, nm + ".conv3", nm + ".bn3", 1, 0, 1, residual=idt, may_split=True)
            outs.append(x)
        return self._heads(N, outs[1], outs[2], outs[3])

    def _heads(self, N, C3, C4, C5):
        l5 = self._conv_b(C5, "fpn.lat_layers.0", 1, 0, 0)
        l4 = self._conv_b(C4, "fpn.lat_layers.1", 1, 0, 0)
        l3 = self._conv_b(C3, "fpn.lat_layers.2", 1, 0, 0)
        x4f = self._h(ora.resize_bilinear(l5, l4.shape[1], l4.shape[2], add=l4))
        x3f = self._h(ora.resize_bilinear(x4f, l3.shape[1], l3.shape[2], add=l3))
        P5 = self._conv_b(l5, "fpn.pred_layers.0", 1, 1, 1)
        P4 = self._conv_b(x4f, "fpn.pred_layers.1", 1, 1, 1)
        P3 = self._conv_b(x3f, "fpn.pred_layers.2", 1, 1, 1)
        P6 = self._conv_b(P5, "fpn.downsample_layers.0", 2, 1, 0)
        P7 = self._conv_b(P6, "fpn.downsample_layers.1", 2, 1, 0)
        P = [P3, P4, P5, P6, P7]
        t = self._conv_b(P3, "proto_net.0", 1, 1, 1)
        t = self._conv_b(t, "proto_net.2", 1, 1, 1)
        t = self._conv_b(t, "proto_net.4", 1, 1, 1)
        t = self._h(ora.resize_bilinear(t, t.shape[1] * 2, t.shape[2] * 2, relu=1))
        t = self._conv_b(t, "proto_net.8", 1, 1, 1)
        proto = self._conv_b(t, "proto_net.10", 1, 0, 1, keep_f32=True)
        locs, confs, masks, priors = [], [], [], []
        scales = self.scales
        for l, p in enumerate(P):
            u = self._conv_b(p, "prediction_layers.0.upfeature.0", 1, 1, 1)
            locs.append(self._conv_b(u, "prediction_layers.0.bbox_layer", 1, 1, 0, keep_f32=True).reshape(N, -1, 4))
            confs.append(self._conv_b(u, "prediction_layers.0.conf_layer", 1, 1, 0, keep_f32=True).reshape(N, -1, 81))
            masks.append(self._conv_b(u, "prediction_layers.0.mask_layer", 1, 1, 2, keep_f32=True).reshape(N, -1, 32))
            lv = tuple(scales[l] * 2 ** (j / 3.0) for j in range(self.scales_per_level))
            priors.append(make_priors(p.shape[1], p.shape[2], lv if self.scales_per_level > 1 else scales[l], self.max_size, square=self.square))
        loc = np.concatenate(locs, 1); conf = np.concatenate(confs, 1); mask = np.concatenate(masks, 1)
        priors = np.concatenate(priors, 0)
        self.feats = dict(C3=C3, C4=C4, C5=C5, P3=P3, P4=P4, P5=P5, P6=P6, P7=P7, proto=proto, loc=loc, conf=conf,
                          mask=mask, priors=priors)
        dets = []
        for n in range(N):
            boxes = ora.yolact_decode(loc[n], priors)
            d = ora.yolact_detect(ora.softmax(conf[n]), boxes, mask[n], second_threshold=self.second_threshold)
            d["proto"] = proto[n]
            dets.append(d)
        return dets

    def mask_scores(self, det, score_threshold=0.0):
        """YOLACT++ fast mask re-scoring (use_maskiou, rescore_mask, not rescore_bbox): FastMaskIoUNet on the cropped proto-
        resolution masks -- five 3x3 stride-2 unpadded convs + ReLU, a 1x1 to the 80 classes + ReLU, global max pool --, the
        detection's own class picked, times the box score.  Returns the mask scores of the kept detections."""
        keep = det["score"] > np.float32(score_threshold)
        box, coeff, cls, score = det["box"][keep], det["mask"][keep], det["cls"][keep], det["score"][keep]
        n = len(score)
        if n == 0:
            return np.zeros((0,), np.float32)
        x = ora.yolact_proto_masks(det["proto"], coeff, box)[..., None]
        for i in (0, 2, 4, 6, 8):
            x = self._conv_b(x, "maskiou_net.%d" % i, 2, 0, 1, keep_f32=True)
        x = self._conv_b(x, "maskiou_net.10", 1, 0, 1, keep_f32=True)
        p = x.reshape(n, -1, x.shape[-1]).max(1)
        return (score * p[np.arange(n), cls]).astype(np.float32)

    @staticmethod
    def postprocess(det, w, h, score_threshold=0.0):
        keep = det["score"] > np.float32(score_threshold)
        box, coeff = det["box"][keep], det["mask"][keep]
        masks, ib = ora.yolact_masks(det["proto"], coeff, box, h, w)
        return det["cls"][keep], det["score"][keep], ib, masks
