"""Composition-level second opinion for the two oracle models (CPU only; parity stays UNPINNED -- the reference tree holds no code -- but
this is the largest surface the op-level cross-checks of test_second_opinion_cpu.py leave unchecked).

The graphs are restated a SECOND time, independently of oracle/{maskrcnn,yolact}_ref.py and of the engine: as torch `nn.Module` trees
shaped and NAMED like the upstream projects the reference names as its sources (README.md:353-358: facebookresearch/maskrcnn-benchmark's
GeneralizedRCNN -- backbone.body.{stem,layer1..4}, backbone.fpn.fpn_{inner,layer}1..4, rpn.head.*, roi_heads.{box,mask}.* --, and
dbolya/yolact's Yolact -- backbone.layers.*, fpn.{lat,pred,downsample}_layers.*, proto_net.*, prediction_layers.0.*), NCHW, torch's own
conv / pooling / interpolation kernels, loaded from the state dict `tools/import_pth.py` produces with `load_state_dict(strict=True)`.
A shared misunderstanding of the COMPOSITION between oracle and engine -- stride in the first 1x1 (Mask R-CNN) vs on the 3x3 (Yolact), FPN
merge order, nearest vs bilinear top-down, P6 from P5, anchor / prior (y, x, a) flattening, head permutes, the FC6 (c, h, w) flatten, the
deconvolution's parity mapping -- would show up here as a mismatch of intermediate tensors.  Tolerance: 1e-4 relative to each tensor's
largest magnitude (the two sides sum in different orders); selections are compared where scores are not tied.
"""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
nn = torch.nn
F = torch.nn.functional


def _close(got_nchw, want_nhwc, tol=1e-4, name=""):
    g = got_nchw.detach().permute(0, 2, 3, 1).numpy() if got_nchw.dim() == 4 else got_nchw.detach().numpy()
    w = np.asarray(want_nhwc)
    assert g.shape == w.shape, (name, g.shape, w.shape)
    err = np.max(np.abs(g - w)) / max(1e-6, np.max(np.abs(w)))
    assert err < tol, (name, err)
    return err


# ------------------------------------------------------------------------------------------------------------------- Mask R-CNN
class FrozenBatchNorm2d(nn.Module):
    """layers/batch_norm.py: fixed statistics and affine parameters, no eps"""

    def __init__(self, n):
        super().__init__()
        for k in ("weight", "bias", "running_mean", "running_var"):
            self.register_buffer(k, torch.zeros(n))

    def forward(self, x):
        scale = self.weight * self.running_var.rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


class Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, stride, stride_in_1x1):
        super().__init__()
        self.downsample = None
        if cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), FrozenBatchNorm2d(cout))
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = nn.Conv2d(cin, mid, 1, stride=s1, bias=False); self.bn1 = FrozenBatchNorm2d(mid)
        self.conv2 = nn.Conv2d(mid, mid, 3, stride=s3, padding=1, bias=False); self.bn2 = FrozenBatchNorm2d(mid)
        self.conv3 = nn.Conv2d(mid, cout, 1, bias=False); self.bn3 = FrozenBatchNorm2d(cout)

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        return F.relu(self.bn3(self.conv3(out)) + idt)


class Stem(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False); self.bn1 = FrozenBatchNorm2d(64)

    def forward(self, x):
        return F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)


class ResNetBody(nn.Module):
    def __init__(self, blocks=(3, 4, 6, 3)):
        super().__init__()
        self.stem = Stem()
        cin = 64
        for li, nb in enumerate(blocks, 1):
            mid, cout = 64 * 2 ** (li - 1), 256 * 2 ** (li - 1)
            layer = [Bottleneck(cin if b == 0 else cout, mid, cout, (2 if li > 1 else 1) if b == 0 else 1, True) for b in range(nb)]
            setattr(self, "layer%d" % li, nn.Sequential(*layer))
            cin = cout

    def forward(self, x):
        x = self.stem(x)
        outs = []
        for li in range(1, 5):
            x = getattr(self, "layer%d" % li)(x)
            outs.append(x)
        return outs


class FPN(nn.Module):
    def __init__(self):
        super().__init__()
        for i, c in enumerate((256, 512, 1024, 2048), 1):
            setattr(self, "fpn_inner%d" % i, nn.Conv2d(c, 256, 1))
            setattr(self, "fpn_layer%d" % i, nn.Conv2d(256, 256, 3, padding=1))

    def forward(self, cs):
        last = self.fpn_inner4(cs[3])
        results = [self.fpn_layer4(last)]
        for i in (3, 2, 1):
            top = F.interpolate(last, scale_factor=2, mode="nearest")
            last = getattr(self, "fpn_inner%d" % i)(cs[i - 1]) + top
            results.insert(0, getattr(self, "fpn_layer%d" % i)(last))
        results.append(F.max_pool2d(results[-1], 1, 2, 0))   # LastLevelMaxPool
        return results


class Backbone(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.body = ResNetBody(blocks); self.fpn = FPN()


class RPNHead(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(256, 256, 3, padding=1); self.cls_logits = nn.Conv2d(256, 3, 1); self.bbox_pred = nn.Conv2d(256, 12, 1)

    def forward(self, feats):
        out = []
        for f in feats:
            t = F.relu(self.conv(f))
            out.append((self.cls_logits(t), self.bbox_pred(t)))
        return out


class _Holder(nn.Module):
    pass


class GeneralizedRCNN(nn.Module):
    def __init__(self, blocks=(3, 4, 6, 3)):
        super().__init__()
        self.backbone = Backbone(blocks)
        self.rpn = _Holder(); self.rpn.head = RPNHead()
        self.roi_heads = _Holder()
        box = _Holder(); box.feature_extractor = _Holder(); box.predictor = _Holder()
        box.feature_extractor.fc6 = nn.Linear(256 * 7 * 7, 1024); box.feature_extractor.fc7 = nn.Linear(1024, 1024)
        box.predictor.cls_score = nn.Linear(1024, 81); box.predictor.bbox_pred = nn.Linear(1024, 324)
        mask = _Holder(); mask.feature_extractor = _Holder(); mask.predictor = _Holder()
        for i in range(1, 5):
            setattr(mask.feature_extractor, "mask_fcn%d" % i, nn.Conv2d(256, 256, 3, padding=1))
        mask.predictor.conv5_mask = nn.ConvTranspose2d(256, 256, 2, 2); mask.predictor.mask_fcn_logits = nn.Conv2d(256, 81, 1)
        self.roi_heads.box, self.roi_heads.mask = box, mask


def _torch_sd(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def _t_roi_align(feat_chw, rois_xyxy, scale, P, g=2):
    """ROIAlign_cpu.cpp (legacy, aligned=False) restated with torch scalar indexing over a small number of RoIs (independent of ora_roi_align)."""
    C, H, W = feat_chw.shape
    out = torch.zeros((len(rois_xyxy), C, P, P))
    for r, (x1, y1, x2, y2) in enumerate(rois_xyxy.tolist()):
        sw, sh, ew, eh = x1 * scale, y1 * scale, x2 * scale, y2 * scale
        rw, rh = max(ew - sw, 1.0), max(eh - sh, 1.0)
        bw, bh = rw / P, rh / P
        for ph in range(P):
            for pw in range(P):
                acc = torch.zeros(C)
                for iy in range(g):
                    y = sh + ph * bh + (iy + 0.5) * bh / g
                    for ix in range(g):
                        x = sw + pw * bw + (ix + 0.5) * bw / g
                        if y < -1.0 or y > H or x < -1.0 or x > W:
                            continue
                        yy, xx = max(y, 0.0), max(x, 0.0)
                        yl, xl = int(yy), int(xx)
                        if yl >= H - 1:
                            yl = yh = H - 1; yy = float(yl)
                        else:
                            yh = yl + 1
                        if xl >= W - 1:
                            xl = xh = W - 1; xx = float(xl)
                        else:
                            xh = xl + 1
                        ly, lx = yy - yl, xx - xl
                        hy, hx = 1 - ly, 1 - lx
                        acc = acc + hy * hx * feat_chw[:, yl, xl] + hy * lx * feat_chw[:, yl, xh] + ly * hx * feat_chw[:, yh, xl] + ly * lx * feat_chw[:, yh, xh]
                out[r, :, ph, pw] = acc / (g * g)
    return out


def _level(box):
    s = math.sqrt((box[2] - box[0] + 1) * (box[3] - box[1] + 1))
    return int(min(5, max(2, math.floor(4 + math.log2(s / 224 + 1e-6)))))


@pytest.mark.parametrize("depth", [50, 101])
def test_maskrcnn_graph_against_module_shaped_restatement(depth):
    from isegmi.maskrcnn import prepare_images
    from isegmi.weights import maskrcnn_state_dict
    from oracle.maskrcnn_ref import MaskRCNNRef, cell_anchors, grid_anchors
    sd = maskrcnn_state_dict(1234, depth)
    model = GeneralizedRCNN((3, 4, 23 if depth == 101 else 6, 3))
    missing = model.load_state_dict(_torch_sd(sd), strict=True)   # every upstream key has a home in the module tree, and vice versa
    assert not missing.missing_keys and not missing.unexpected_keys
    model.eval()
    rng = np.random.default_rng(3)
    imgs = [rng.uniform(0, 255, (100, 130, 3)).astype(np.float32), rng.uniform(0, 255, (96, 120, 3)).astype(np.float32)]
    x, hw = prepare_images(imgs)
    ref = MaskRCNNRef(sd, depth=depth)
    dets = ref.forward(x, hw)
    with torch.no_grad():
        xin = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
        cs = model.backbone.body(xin)
        ps = model.backbone.fpn(cs)
        rpn = model.rpn.head(ps)
        _close(cs[0], ref.feats["C2"], name="C2"); _close(cs[3], ref.feats["C5"], name="C5")
        for l, nm in enumerate(("P2", "P3", "P4", "P5", "P6")):
            _close(ps[l], ref.feats[nm], name=nm)
        # RPN: objectness permuted N,A,H,W -> N,H,W,A; deltas N,4A,H,W -> N,H,W,A,4; anchors enumerated (y, x, a)
        for l, (lg, dl) in enumerate(rpn):
            _close(lg, ref.dbg["rpn_logits"][l], name="rpn logits %d" % l)
            _close(dl, ref.dbg["rpn_deltas"][l], name="rpn deltas %d" % l)
        # upstream's flattening + anchor order on the finest level: the oracle's best proposal of level P2 must decode from the anchor /
        # delta pair the permute-and-flatten convention points at
        lg, dl = rpn[0]
        N, A, H, W = lg.shape
        flat_obj = lg.permute(0, 2, 3, 1).reshape(N, -1)
        flat_del = dl.view(N, A, 4, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, 4)
        anc = torch.from_numpy(grid_anchors(H, W, 4, cell_anchors(4, 32)))
        from oracle import ora
        for n in range(N):
            lb, ls = ora.rpn_level(ref.dbg["rpn_logits"][0][n].reshape(-1), ref.dbg["rpn_deltas"][0][n].reshape(-1, 4), anc.numpy(), 1000, 1000, 0.7, 0.0,
                                   float(hw[n][1]), float(hw[n][0]))
            prob = torch.sigmoid(flat_obj[n]).numpy()
            assert abs(float(prob.max()) - float(ls[0])) < 1e-5          # the first kept proposal carries the level's best objectness ...
            cand = np.nonzero(np.abs(prob - ls[0]) < 1e-6)[0][:64]        # ... and is the decode of an anchor with that objectness (saturated scores tie)
            dec = ora.decode_boxes(anc[cand].numpy(), flat_del[n, cand].numpy(), (1.0, 1.0, 1.0, 1.0), float(hw[n][1]), float(hw[n][0]))
            assert np.min(np.max(np.abs(dec - lb[0][None]), 1)) < 1e-3
        # box head on the oracle's proposals: Pooler (LevelMapper + legacy RoIAlign) -> flatten (C, H, W) -> fc6 -> fc7 -> predictors
        n = 0
        props = dets[n]["proposals"][:6]
        feats = torch.stack([_t_roi_align(ps[_level(b) - 2][n], torch.from_numpy(b[None]), 1.0 / (4 * 2 ** (_level(b) - 2)), 7)[0] for b in props])
        f6 = F.relu(model.roi_heads.box.feature_extractor.fc6(feats.reshape(len(props), -1)))
        f7 = F.relu(model.roi_heads.box.feature_extractor.fc7(f6))
        cls, reg = model.roi_heads.box.predictor.cls_score(f7), model.roi_heads.box.predictor.bbox_pred(f7)
        _close(cls, ref.dbg["cls"][n][:6], name="cls_score"); _close(reg, ref.dbg["reg"][n][:6], name="bbox_pred")
        # mask head on the oracle's detections: RoIAlign 14x14 -> 4 x (conv3x3 + ReLU) -> ConvTranspose2d(2, 2) + ReLU -> 1x1 -> sigmoid -> label's channel
        k = min(4, len(dets[n]["box"]))
        assert k > 0
        db, dl_ = dets[n]["box"][:k], dets[n]["label"][:k]
        mf = torch.stack([_t_roi_align(ps[_level(b) - 2][n], torch.from_numpy(b[None]), 1.0 / (4 * 2 ** (_level(b) - 2)), 14)[0] for b in db])
        for i in range(1, 5):
            mf = F.relu(getattr(model.roi_heads.mask.feature_extractor, "mask_fcn%d" % i)(mf))
        up = F.relu(model.roi_heads.mask.predictor.conv5_mask(mf))
        prob = torch.sigmoid(model.roi_heads.mask.predictor.mask_fcn_logits(up))
        m28 = prob[torch.arange(k), torch.from_numpy(dl_.astype(np.int64))]
        assert np.max(np.abs(m28.numpy() - dets[n]["mask28"][:k])) < 1e-4


# ------------------------------------------------------------------------------------------------------------------------ Yolact
class YBottleneck(nn.Module):
    """yolact backbone.py Bottleneck: torchvision-style, stride on the 3x3, BatchNorm2d(eval, eps 1e-5)"""

    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False); self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        return F.relu(self.bn3(self.conv3(out)) + idt)


class YResNet(nn.Module):
    def __init__(self, blocks=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False); self.bn1 = nn.BatchNorm2d(64)
        self.layers = nn.ModuleList()
        cin = 64
        for li, nb in enumerate(blocks):
            planes, stride = 64 * 2 ** li, (1 if li == 0 else 2)
            ds = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
            layer = [YBottleneck(cin, planes, stride, ds)] + [YBottleneck(planes * 4, planes, 1, None) for _ in range(nb - 1)]
            self.layers.append(nn.Sequential(*layer))
            cin = planes * 4

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        outs = []
        for layer in self.layers:
            x = layer(x)
            outs.append(x)
        return outs


class YFPN(nn.Module):
    def __init__(self):
        super().__init__()
        self.lat_layers = nn.ModuleList([nn.Conv2d(c, 256, 1) for c in (2048, 1024, 512)])
        self.pred_layers = nn.ModuleList([nn.Conv2d(256, 256, 3, padding=1) for _ in range(3)])
        self.downsample_layers = nn.ModuleList([nn.Conv2d(256, 256, 3, stride=2, padding=1) for _ in range(2)])

    def forward(self, convouts):   # C3, C4, C5
        out = [None, None, None]
        x = torch.zeros(1)
        j = 3
        for lat in self.lat_layers:
            j -= 1
            if j < 2:
                _, _, h, w = convouts[j].shape
                x = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)
            x = x + lat(convouts[j])
            out[j] = x
        j = 3
        for pred in self.pred_layers:
            j -= 1
            out[j] = F.relu(pred(out[j]))
        for ds in self.downsample_layers:
            out.append(ds(out[-1]))
        return out


class PredictionModule(nn.Module):
    def __init__(self, A=3):
        super().__init__()
        self.upfeature = nn.Sequential(nn.Conv2d(256, 256, 3, padding=1), nn.ReLU())
        self.bbox_layer = nn.Conv2d(256, A * 4, 3, padding=1)
        self.conf_layer = nn.Conv2d(256, A * 81, 3, padding=1)
        self.mask_layer = nn.Conv2d(256, A * 32, 3, padding=1)

    def forward(self, x):
        x = self.upfeature(x)
        n = x.shape[0]
        bbox = self.bbox_layer(x).permute(0, 2, 3, 1).contiguous().view(n, -1, 4)
        conf = self.conf_layer(x).permute(0, 2, 3, 1).contiguous().view(n, -1, 81)
        mask = torch.tanh(self.mask_layer(x).permute(0, 2, 3, 1).contiguous().view(n, -1, 32))
        return bbox, conf, mask


class YolactNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.backbone = YResNet()
        self.fpn = YFPN()
        # make_net(mask_proto_net): [(256,3,p1)] x 3 + [(None,-2)] + [(256,3,p1)] + [(32,1)]; indices 0 2 4 (6 = interpolate) 8 10, ReLU after every conv but the last
        pn = [nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(), nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(), nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(),
              nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False), nn.ReLU(), nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(), nn.Conv2d(256, 32, 1)]
        self.proto_net = nn.Sequential(*pn)
        self.prediction_layers = nn.ModuleList([PredictionModule()])   # share_prediction_module

    def forward(self, x):
        outs = self.backbone(x)
        fo = self.fpn([outs[1], outs[2], outs[3]])
        proto = F.relu(self.proto_net(fo[0]))   # mask_proto_prototype_activation = relu (upstream then permutes to NHWC; _close does)
        preds = [self.prediction_layers[0](f) for f in fo]
        return outs, fo, proto, [torch.cat([p[i] for p in preds], 1) for i in range(3)]


def test_yolact_graph_against_module_shaped_restatement():
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import fast_base_transform
    from oracle.yolact_ref import YolactRef
    sd = yolact_state_dict(1234)
    tsd = _torch_sd(sd)
    for k in list(sd):
        if k.endswith("running_var"):
            tsd[k[:-11] + "num_batches_tracked"] = torch.tensor(0)   # nn.BatchNorm2d's counter: present in real yolact .pth files, dropped by the importer
    model = YolactNet()
    res = model.load_state_dict(tsd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model.eval()
    rng = np.random.default_rng(5)
    x = fast_base_transform(rng.uniform(0, 255, (2, 140, 140, 3)).astype(np.float32))
    ref = YolactRef(sd)
    dets = ref.forward(x)
    with torch.no_grad():
        outs, fo, proto, (loc, conf, mask) = model(torch.from_numpy(x).permute(0, 3, 1, 2).contiguous())
    for t, nm in ((outs[1], "C3"), (outs[2], "C4"), (outs[3], "C5"), (fo[0], "P3"), (fo[1], "P4"), (fo[2], "P5"), (fo[3], "P6"), (fo[4], "P7")):
        _close(t, ref.feats[nm], name=nm)
    _close(proto, ref.feats["proto"], name="proto")
    _close(loc, ref.feats["loc"], name="loc"); _close(conf, ref.feats["conf"], name="conf"); _close(mask, ref.feats["mask"], name="mask")
    # Detect on the module's own outputs with the tensor-op restatement of test_second_opinion_cpu.py: same detections where scores are untied
    from test_second_opinion_cpu import t_decode, t_detect
    pri = torch.from_numpy(ref.feats["priors"])
    for n in range(2):
        boxes = t_decode(loc[n], pri)
        d = t_detect(F.softmax(conf[n], -1), boxes, mask[n])
        r = dets[n]
        k = min(len(d["score"]), len(r["score"]), 20)
        assert k > 0
        gaps = np.abs(np.diff(r["score"][:k + 1])) if len(r["score"]) > k else np.ones(k)
        for i in range(k):
            if i < len(gaps) and gaps[i] < 1e-5 or (i > 0 and gaps[i - 1] < 1e-5):
                continue   # a near-tie may order differently under 1e-6 score noise
            assert int(d["cls"][i]) == int(r["cls"][i]) and abs(float(d["score"][i]) - float(r["score"][i])) < 1e-4
            assert np.max(np.abs(d["box"][i].numpy() - r["box"][i])) < 1e-4
