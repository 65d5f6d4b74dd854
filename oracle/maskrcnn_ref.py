"""CPU oracle of the Mask R-CNN R50/R101-FPN forward (TEST INFRASTRUCTURE ONLY).

numpy graph over the C oracle ops.  Follows SURVEY.md Appendix A.2-A.8 ([UPSTREAM-RECALL] of
facebookresearch/maskrcnn-benchmark, the lineage the reference names at README.md:358; the
reference's detectron.jittor sources are absent -> PARITY UNPINNED).  Consumes the upstream-named
state dict (OIHW conv weights, FrozenBN stats, [out,in] FC weights, [in,out,2,2] deconv) as is.
"""
import numpy as np

from . import ora


def _krsc(w):
    return np.ascontiguousarray(np.transpose(np.asarray(w, np.float32), (0, 2, 3, 1)))


def _frozen_bn(sd, p, eps=0.0):
    """FrozenBatchNorm2d (A.1): scale = w * rsqrt(var) (no eps); shift = b - mean*scale.  eps > 0: the A.1 fork, rsqrt(var + eps)."""
    w, b, m, v = (sd[p + k].astype(np.float32) for k in (".weight", ".bias", ".running_mean", ".running_var"))
    if eps:
        v = (v + np.float32(eps)).astype(np.float32)
    scale = (w * (np.float32(1.0) / np.sqrt(v))).astype(np.float32)
    return scale, (b - m * scale).astype(np.float32)


def cell_anchors(stride, size, ratios=(0.5, 1.0, 2.0)):
    """A.3 generate_anchors: ratio enumeration with np.round, then scale enumeration."""
    def whctrs(a):
        w = a[2] - a[0] + 1; h = a[3] - a[1] + 1
        return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)

    def mk(ws, hs, xc, yc):
        return np.stack([xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)], -1)
    base = np.array([0, 0, stride - 1, stride - 1], np.float64)
    w, h, xc, yc = whctrs(base)
    out = []
    for r in ratios:
        ws = np.round(np.sqrt(w * h / r)); hs = np.round(ws * r)
        a = mk(ws, hs, xc, yc)
        w2, h2, xc2, yc2 = whctrs(a)
        out.append(mk(w2 * (size / stride), h2 * (size / stride), xc2, yc2))
    return np.asarray(out, np.float64).astype(np.float32)


def cell_anchors_multi(stride, sizes, ratios=(0.5, 1.0, 2.0)):
    """Single-map form: for each ratio (outer) every size (inner) -> 15 anchors at stride 16 for the C4 config."""
    return np.concatenate([np.stack([cell_anchors(stride, s, (r,))[0] for s in sizes]) for r in ratios], 0).astype(np.float32)


def grid_anchors(gh, gw, stride, cell):
    out = np.empty((gh, gw, cell.shape[0], 4), np.float32)
    for y in range(gh):
        for x in range(gw):
            sh = np.array([x * stride, y * stride, x * stride, y * stride], np.float32)
            out[y, x] = sh[None, :] + cell
    return out.reshape(-1, 4)


class MaskRCNNRef:
    def __init__(self, sd, depth=50, pre_nms=1000, post_nms=1000, fpn_post=1000, det_per_img=100, nms_ge=0, fp16=False,
                 nms_plus_one=1, nms_index_order=0, roi_aligned=0, bn_eps=0.0, conv_split_k=0):
        """The keyword forks are SURVEY 7.2 / App. A.1, A.6, A.7 (defaults: maskrcnn-benchmark's CUDA path): nms_ge 1 suppress on iou >= thr;
        nms_plus_one 0 plain areas in the NMS IoU; nms_index_order 1 a class's detections in proposal-index order (CPU NMS); roi_aligned 1
        ROIAlign(aligned=True); bn_eps FrozenBatchNorm2d's rsqrt(var + eps)."""
        self.sd, self.depth = sd, depth
        self.aligned, self.bn_eps = int(roi_aligned), float(bn_eps)
        self.conv_split_k = int(conv_split_k)   # the product's opt-in split-K numerics mode (see YolactRef): conv2 / conv3 and conv1 of later blocks, by the shape rule
        # fp16=True emulates the product's fp16-storage path (BASELINE configs[4]): conv weights, the input image and every
        # stored activation are rounded to fp16, all arithmetic stays fp32 (ordered fmaf chain).
        self.fp16 = fp16
        self.pre_nms, self.post_nms, self.fpn_post, self.dpi = pre_nms, post_nms, fpn_post, det_per_img
        self.ge = (1 if nms_ge else 0) | (0 if nms_plus_one else 2)      # flags of ora.rpn_level
        self.ge_box = self.ge | (4 if nms_index_order else 0)          # flags of ora.box_postprocess
        self.feats = {}

    def _h(self, x):
        return x.astype(np.float16).astype(np.float32) if self.fp16 else x

    def _cbn(self, x, conv, bn, stride, pad, act, residual=None, may_split=False):
        sc, sh = _frozen_bn(self.sd, bn, self.bn_eps)
        w = self._h(_krsc(self.sd[conv + ".weight"]))
        ks = 1
        if may_split and self.conv_split_k and not self.fp16:
            ho, wo = (x.shape[1] + 2 * pad - w.shape[1]) // stride + 1, (x.shape[2] + 2 * pad - w.shape[2]) // stride + 1
            ks = 4 if ora.conv_split_qualifies(x.shape[0] * ho * wo, w.shape[0], w.shape[1], w.shape[2], w.shape[3]) else 1
        return self._h(ora.conv2d(x, w, stride, pad, sc, sh, residual, act, ksplit=ks))

    def _cb(self, x, name, stride, pad, act, keep_f32=False):
        y = ora.conv2d(x, self._h(_krsc(self.sd[name + ".weight"])), stride, pad, None, self.sd[name + ".bias"], None, act)
        return y if keep_f32 else self._h(y)

    def forward(self, images_nhwc3, image_hw):
        sd = self.sd
        x = np.asarray(images_nhwc3, np.float32)
        N = x.shape[0]
        x4 = np.concatenate([x, np.zeros(x.shape[:3] + (1,), np.float32)], -1)
        w1 = _krsc(sd["backbone.body.stem.conv1.weight"])
        w1 = np.concatenate([w1, np.zeros(w1.shape[:3] + (1,), np.float32)], -1)
        sc, sh = _frozen_bn(sd, "backbone.body.stem.bn1", self.bn_eps)
        x = ora.maxpool(self._h(ora.conv2d(self._h(x4), self._h(w1), 2, 3, sc, sh, None, 1)), 3, 2, 1)  # fp16 mode: image, weights, stem output rounded
        Cs = []
        for li, nb in enumerate((3, 4, 23 if self.depth == 101 else 6, 3), 1):
            for b in range(nb):
                nm = "backbone.body.layer%d.%d" % (li, b)
                st = 2 if (b == 0 and li > 1) else 1
                idt = self._cbn(x, nm + ".downsample.0", nm + ".downsample.1", st, 0, 0) if b == 0 else x
                t = self._cbn(x, nm + ".conv1", nm + ".bn1", st, 0, 1, may_split=b > 0)
                t = self._cbn(t, nm + ".conv2", nm + ".bn2", 1, 1, 1, may_split=True)
                x = self._cbn(t, nm + ".conv3", nm + ".bn3", 1, 0, 1, residual=idt, may_split=True)
            Cs.append(x)
        last = self._cb(Cs[3], "backbone.fpn.fpn_inner4", 1, 0, 0)
        P = [None, None, None, self._cb(last, "backbone.fpn.fpn_layer4", 1, 1, 0)]
        for l in (2, 1, 0):
            lat = self._cb(Cs[l], "backbone.fpn.fpn_inner%d" % (l + 1), 1, 0, 0)
            last = self._h(ora.upsample_nearest2x_add(last, lat))
            P[l] = self._cb(last, "backbone.fpn.fpn_layer%d" % (l + 1), 1, 1, 0)
        P.append(ora.maxpool(P[3], 1, 2, 0))
        # RPN
        strides, sizes = (4, 8, 16, 32, 64), (32, 64, 128, 256, 512)
        props, pscores = [], []
        lvl_out = [[] for _ in range(N)]
        self.dbg = dict(rpn_logits=[], rpn_deltas=[], cls=[], reg=[], f7=[])
        for l, p in enumerate(P):
            t = self._cb(p, "rpn.head.conv", 1, 1, 1)
            logits = self._cb(t, "rpn.head.cls_logits", 1, 0, 0, keep_f32=True)  # [N,H,W,A]
            deltas = self._cb(t, "rpn.head.bbox_pred", 1, 0, 0, keep_f32=True)   # [N,H,W,A*4]
            anc = grid_anchors(p.shape[1], p.shape[2], strides[l], cell_anchors(strides[l], sizes[l]))
            self.dbg['rpn_logits'].append(logits); self.dbg['rpn_deltas'].append(deltas)
            for n in range(N):
                b, s = ora.rpn_level(logits[n].reshape(-1), deltas[n].reshape(-1, 4), anc, self.pre_nms, self.post_nms, 0.7, 0.0,
                                     float(image_hw[n][1]), float(image_hw[n][0]), self.ge)
                lvl_out[n].append((b, s))
        for n in range(N):
            b = np.concatenate([q[0] for q in lvl_out[n]], 0); s = np.concatenate([q[1] for q in lvl_out[n]], 0)
            ts, ti = ora.topk(s, min(self.fpn_post, len(s)))
            props.append(b[ti]); pscores.append(ts)
        # box head
        dets = []
        w6 = sd["roi_heads.box.feature_extractor.fc6.weight"].astype(np.float32)
        w7 = sd["roi_heads.box.feature_extractor.fc7.weight"].astype(np.float32)
        for n in range(N):
            pr = props[n]
            R = pr.shape[0]
            lv = ora.level_map(pr)
            feat = np.zeros((R, 7, 7, 256), np.float32)
            for k in range(2, 6):
                idx = np.nonzero(lv == k)[0]
                if len(idx) == 0:
                    continue
                rois = np.concatenate([np.full((len(idx), 1), n, np.float32), pr[idx]], 1)
                feat[idx] = self._h(ora.roi_align(P[k - 2], rois, 1.0 / strides[k - 2], 7, 7, 2, self.aligned))
            # FC6 on the flattened (C,H,W) vector == 7x7 valid conv with weights permuted to (H,W,C)
            w6k = np.ascontiguousarray(w6.reshape(1024, 256, 7, 7).transpose(0, 2, 3, 1))
            f6 = self._h(ora.conv2d(feat, self._h(w6k), 1, 0, None, sd["roi_heads.box.feature_extractor.fc6.bias"], None, 1))
            f7 = self._h(ora.conv2d(f6, self._h(w7.reshape(1024, 1, 1, 1024)), 1, 0, None, sd["roi_heads.box.feature_extractor.fc7.bias"], None, 1))
            cls = ora.conv2d(f7, self._h(sd["roi_heads.box.predictor.cls_score.weight"].reshape(81, 1, 1, 1024)), 1, 0, None,
                             sd["roi_heads.box.predictor.cls_score.bias"], None, 0).reshape(R, 81)
            reg = ora.conv2d(f7, self._h(sd["roi_heads.box.predictor.bbox_pred.weight"].reshape(324, 1, 1, 1024)), 1, 0, None,
                             sd["roi_heads.box.predictor.bbox_pred.bias"], None, 0).reshape(R, 324)
            self.dbg['cls'].append(cls); self.dbg['reg'].append(reg); self.dbg['f7'].append(f7)
            db, ds, dl = ora.box_postprocess(cls, reg, pr, float(image_hw[n][1]), float(image_hw[n][0]), 0.05, 0.5, self.dpi, self.ge_box, self.dpi)
            # mask head
            D = db.shape[0]
            m28 = np.zeros((D, 28, 28), np.float32)
            if D:
                lv = ora.level_map(db)
                mf = np.zeros((D, 14, 14, 256), np.float32)
                for k in range(2, 6):
                    idx = np.nonzero(lv == k)[0]
                    if len(idx) == 0:
                        continue
                    rois = np.concatenate([np.full((len(idx), 1), n, np.float32), db[idx]], 1)
                    mf[idx] = self._h(ora.roi_align(P[k - 2], rois, 1.0 / strides[k - 2], 14, 14, 2, self.aligned))
                for i in range(1, 5):
                    mf = self._cb(mf, "roi_heads.mask.feature_extractor.mask_fcn%d" % i, 1, 1, 1)
                up = self._h(ora.deconv2x2(mf, self._h(sd["roi_heads.mask.predictor.conv5_mask.weight"].astype(np.float32)),
                                           sd["roi_heads.mask.predictor.conv5_mask.bias"], 1))
                m28 = ora.mask_logits_select(up.reshape(D, 784, 256), sd["roi_heads.mask.predictor.mask_fcn_logits.weight"].reshape(81, 256),
                                             sd["roi_heads.mask.predictor.mask_fcn_logits.bias"], dl).reshape(D, 28, 28)
            dets.append(dict(box=db, score=ds, label=dl, mask28=m28, proposals=pr, proposal_scores=pscores[n]))
        self.feats = dict(C2=Cs[0], C5=Cs[3], P2=P[0], P3=P[1], P4=P[2], P5=P[3], P6=P[4])
        return dets

    # ---- e2e_mask_rcnn_R_50_C4_1x (README.md:263-273): one stride-16 map, conv5 head shared by box and mask branches
    def _res5(self, x):
        sd = self.sd
        for b in range(3):
            nm = "roi_heads.box.feature_extractor.head.layer4.%d" % b
            st = 2 if b == 0 else 1
            idt = self._cbn(x, nm + ".downsample.0", nm + ".downsample.1", st, 0, 0) if b == 0 else x
            t = self._cbn(x, nm + ".conv1", nm + ".bn1", st, 0, 1)
            t = self._cbn(t, nm + ".conv2", nm + ".bn2", 1, 1, 1)
            x = self._cbn(t, nm + ".conv3", nm + ".bn3", 1, 0, 1, residual=idt)
        return x

    def forward_c4(self, images_nhwc3, image_hw, pre_nms=6000, post_nms=1000):
        """App. A restated for the C4 config: anchors 5 sizes x 3 ratios (ratio-major) at stride 16; top pre_nms -> decode /
        clip / NMS 0.7 -> post_nms, no cross-level merge; ROIAlign 14x14 scale 1/16 sampling_ratio 0; conv5 head (first
        block stride 2) -> AvgPool 7 -> cls_score | bbox_pred; same box post-processing; the shared extractor on the
        detections -> ConvTranspose 2x2/2 2048->256 + ReLU -> 1x1 -> class-selected sigmoid: 14x14 masks."""
        assert not self.fp16
        sd = self.sd
        x = np.asarray(images_nhwc3, np.float32)
        N = x.shape[0]
        x4 = np.concatenate([x, np.zeros(x.shape[:3] + (1,), np.float32)], -1)
        w1 = _krsc(sd["backbone.body.stem.conv1.weight"])
        w1 = np.concatenate([w1, np.zeros(w1.shape[:3] + (1,), np.float32)], -1)
        sc, sh = _frozen_bn(sd, "backbone.body.stem.bn1", self.bn_eps)
        x = ora.maxpool(ora.conv2d(x4, w1, 2, 3, sc, sh, None, 1), 3, 2, 1)
        for li, nb in enumerate((3, 4, 6), 1):
            for b in range(nb):
                nm = "backbone.body.layer%d.%d" % (li, b)
                st = 2 if (b == 0 and li > 1) else 1
                idt = self._cbn(x, nm + ".downsample.0", nm + ".downsample.1", st, 0, 0) if b == 0 else x
                t = self._cbn(x, nm + ".conv1", nm + ".bn1", st, 0, 1)
                t = self._cbn(t, nm + ".conv2", nm + ".bn2", 1, 1, 1)
                x = self._cbn(t, nm + ".conv3", nm + ".bn3", 1, 0, 1, residual=idt)
        C4 = x
        t = self._cb(C4, "rpn.head.conv", 1, 1, 1)
        logits = self._cb(t, "rpn.head.cls_logits", 1, 0, 0, keep_f32=True)   # [N,H,W,15]
        deltas = self._cb(t, "rpn.head.bbox_pred", 1, 0, 0, keep_f32=True)    # [N,H,W,60]
        anc = grid_anchors(C4.shape[1], C4.shape[2], 16, cell_anchors_multi(16, (32, 64, 128, 256, 512)))
        dets = []
        for n in range(N):
            pr, ps = ora.rpn_level(logits[n].reshape(-1), deltas[n].reshape(-1, 4), anc, pre_nms, post_nms, 0.7, 0.0,
                                   float(image_hw[n][1]), float(image_hw[n][0]), self.ge)
            R = pr.shape[0]
            rois = np.concatenate([np.full((R, 1), n, np.float32), pr], 1)
            f5 = self._res5(ora.roi_align(C4, rois, 1.0 / 16, 14, 14, 0, self.aligned))
            pooled = ora.avgpool_full(f5).reshape(R, 1, 1, -1)
            cls = ora.conv2d(pooled, sd["roi_heads.box.predictor.cls_score.weight"].reshape(81, 1, 1, -1), 1, 0, None,
                             sd["roi_heads.box.predictor.cls_score.bias"], None, 0).reshape(R, 81)
            reg = ora.conv2d(pooled, sd["roi_heads.box.predictor.bbox_pred.weight"].reshape(324, 1, 1, -1), 1, 0, None,
                             sd["roi_heads.box.predictor.bbox_pred.bias"], None, 0).reshape(R, 324)
            db, ds, dl = ora.box_postprocess(cls, reg, pr, float(image_hw[n][1]), float(image_hw[n][0]), 0.05, 0.5, self.dpi, self.ge_box, self.dpi)
            D = db.shape[0]
            m14 = np.zeros((D, 14, 14), np.float32)
            if D:
                mr = np.concatenate([np.full((D, 1), n, np.float32), db], 1)
                m5 = self._res5(ora.roi_align(C4, mr, 1.0 / 16, 14, 14, 0, self.aligned))
                up = ora.deconv2x2(m5, sd["roi_heads.mask.predictor.conv5_mask.weight"].astype(np.float32),
                                   sd["roi_heads.mask.predictor.conv5_mask.bias"], 1)
                m14 = ora.mask_logits_select(up.reshape(D, 196, 256), sd["roi_heads.mask.predictor.mask_fcn_logits.weight"].reshape(81, 256),
                                             sd["roi_heads.mask.predictor.mask_fcn_logits.bias"], dl).reshape(D, 14, 14)
            dets.append(dict(box=db, score=ds, label=dl, mask28=m14, proposals=pr, proposal_scores=ps, cls=cls))
        self.feats = dict(C4=C4)
        return dets

    @staticmethod
    def paste(det, out_h, out_w, ratio_wh=(1.0, 1.0), thr=0.5):
        r = np.array([ratio_wh[0], ratio_wh[1], ratio_wh[0], ratio_wh[1]], np.float32)
        boxes = (det["box"] * r).astype(np.float32)
        return ora.paste_masks(det["mask28"], boxes, out_h, out_w, thr), boxes
