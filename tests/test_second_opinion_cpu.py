"""Independent second opinion on the oracle's selection / assembly ops (tests only; never imported by the product).

Parity against the Jittor reference cannot be pinned (its submodules are empty: SURVEY.md 8c), and the HIP kernels and the C
oracle were written from the same recalled specification.  To shrink the shared-misconception risk, every op below is restated
a SECOND time with plain torch-CPU tensor ops, in the tensor-op shape the named lineage uses (dbolya/yolact
layers/functions/detection.py + layers/box_utils.py + layers/output_utils.py; maskrcnn-benchmark modeling/box_coder.py,
modeling/rpn/inference.py, modeling/roi_heads/box_head/inference.py, modeling/roi_heads/mask_head/inference.py) -- sort ->
pairwise IoU -> triu(diagonal=1) -> column max <= thr, softmax, decode with variances, sanitize / crop, kthvalue cut, the
greedy NMS loop -- and compared with `oracle.ora`: exact agreement of every index / label / count, <= 1e-6 on floats.
Inputs are seeded and include exact score ties and IoU == threshold cases."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from oracle import ora  # noqa: E402

TOL = 1e-6


# ------------------------------------------------------------------------------------------------ Yolact (dbolya/yolact shape)
def t_decode(loc, priors, variances=(0.1, 0.2)):
    boxes = torch.cat((priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:],
                       priors[:, 2:] * torch.exp(loc[:, 2:] * variances[1])), 1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def t_intersect(box_a, box_b):
    n, A, B = box_a.size(0), box_a.size(1), box_b.size(1)
    max_xy = torch.min(box_a[:, :, 2:].unsqueeze(2).expand(n, A, B, 2), box_b[:, :, 2:].unsqueeze(1).expand(n, A, B, 2))
    min_xy = torch.max(box_a[:, :, :2].unsqueeze(2).expand(n, A, B, 2), box_b[:, :, :2].unsqueeze(1).expand(n, A, B, 2))
    inter = torch.clamp(max_xy - min_xy, min=0)
    return inter[:, :, :, 0] * inter[:, :, :, 1]


def t_jaccard(box_a, box_b):
    inter = t_intersect(box_a, box_b)
    area_a = ((box_a[:, :, 2] - box_a[:, :, 0]) * (box_a[:, :, 3] - box_a[:, :, 1])).unsqueeze(2).expand_as(inter)
    area_b = ((box_b[:, :, 2] - box_b[:, :, 0]) * (box_b[:, :, 3] - box_b[:, :, 1])).unsqueeze(1).expand_as(inter)
    union = area_a + area_b - inter
    return inter / union


def t_detect(conf, boxes, masks, conf_thresh=0.05, nms_thresh=0.5, top_k=200, max_det=100):
    """Detect.detect + fast_nms for one image; conf [P, 81] (after softmax), boxes [P, 4], masks [P, 32]."""
    cur_scores = conf.t()[1:, :]
    conf_scores, _ = torch.max(cur_scores, dim=0)
    keep = conf_scores > conf_thresh
    kept_idx = torch.nonzero(keep).flatten()
    scores = cur_scores[:, keep]
    boxes = boxes[keep, :]
    masks = masks[keep, :]
    if scores.size(1) == 0:
        return None
    scores, idx = scores.sort(dim=1, descending=True, stable=True)
    idx = idx[:, :top_k].contiguous()
    scores = scores[:, :top_k]
    num_classes, num_dets = idx.size()
    prior = kept_idx[idx.view(-1)].view(num_classes, num_dets)
    boxes = boxes[idx.view(-1), :].view(num_classes, num_dets, 4)
    masks = masks[idx.view(-1), :].view(num_classes, num_dets, -1)
    iou = t_jaccard(boxes, boxes)
    iou.triu_(diagonal=1)
    iou_max, _ = iou.max(dim=1)
    keep = iou_max <= nms_thresh
    classes = torch.arange(num_classes)[:, None].expand_as(keep)
    classes, boxes, masks, scores, prior = classes[keep], boxes[keep], masks[keep], scores[keep], prior[keep]
    scores, idx = scores.sort(dim=0, descending=True, stable=True)
    idx = idx[:max_det]
    return dict(box=boxes[idx], score=scores[:max_det], cls=classes[idx], mask=masks[idx], prior=prior[idx])


def t_sanitize(_x1, _x2, img_size, padding=0, cast=True):
    _x1 = _x1 * img_size
    _x2 = _x2 * img_size
    if cast:
        _x1 = _x1.long()
        _x2 = _x2.long()
    x1 = torch.min(_x1, _x2)
    x2 = torch.max(_x1, _x2)
    x1 = torch.clamp(x1 - padding, min=0)
    x2 = torch.clamp(x2 + padding, max=img_size)
    return x1, x2


def t_crop(masks, boxes, padding=1):
    h, w, n = masks.size()
    x1, x2 = t_sanitize(boxes[:, 0], boxes[:, 2], w, padding, cast=False)
    y1, y2 = t_sanitize(boxes[:, 1], boxes[:, 3], h, padding, cast=False)
    rows = torch.arange(w, dtype=x1.dtype).view(1, -1, 1).expand(h, w, n)
    cols = torch.arange(h, dtype=x1.dtype).view(-1, 1, 1).expand(h, w, n)
    crop_mask = (rows >= x1.view(1, 1, -1)) * (rows < x2.view(1, 1, -1)) * (cols >= y1.view(1, 1, -1)) * (cols < y2.view(1, 1, -1))
    return masks * crop_mask.float()


def t_postprocess(proto, coeff, boxes, w, h):
    masks = proto @ coeff.t()
    masks = torch.sigmoid(masks)
    masks = t_crop(masks, boxes)
    lo = masks.permute(2, 0, 1).contiguous()
    up = F.interpolate(lo.unsqueeze(0), (h, w), mode="bilinear", align_corners=False).squeeze(0)
    hard = up.gt(0.5)
    b = boxes.clone()
    b[:, 0], b[:, 2] = t_sanitize(boxes[:, 0], boxes[:, 2], w, cast=False)
    b[:, 1], b[:, 3] = t_sanitize(boxes[:, 1], boxes[:, 3], h, cast=False)
    return lo, up, hard, b.long()


def _yolact_inputs(seed, P=600, ties=True):
    rng = np.random.default_rng(seed)
    logits = rng.standard_normal((P, 81)).astype(np.float32)
    logits[:, 0] += 2.5
    hot = rng.integers(0, P, 90)
    logits[hot, rng.integers(1, 8, 90)] += 7.0  # few classes -> crowded per-class lists, heavy overlap
    pri = np.concatenate([rng.uniform(0.2, 0.8, (P, 2)), rng.uniform(0.1, 0.5, (P, 2))], 1).astype(np.float32)
    loc = (rng.standard_normal((P, 4)) * 0.4).astype(np.float32)
    if ties:  # exact duplicates: equal scores AND equal boxes (IoU exactly 1), resolved by index order
        for a, b in ((hot[0], hot[1]), (hot[2], hot[3]), (hot[4], hot[5])):
            logits[b] = logits[a]; pri[b] = pri[a]; loc[b] = loc[a]
    msk = np.tanh(rng.standard_normal((P, 32))).astype(np.float32)
    return logits, pri, loc, msk


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_yolact_softmax_decode_fast_nms_against_torch(seed):
    logits, pri, loc, msk = _yolact_inputs(seed)
    conf = ora.softmax(logits)
    tconf = torch.softmax(torch.from_numpy(logits), -1)
    assert np.abs(conf - tconf.numpy()).max() <= TOL
    boxes = ora.yolact_decode(loc, pri)
    tboxes = t_decode(torch.from_numpy(loc), torch.from_numpy(pri))
    assert np.abs(boxes - tboxes.numpy()).max() <= TOL
    # selection on IDENTICAL inputs (the oracle's conf / boxes): indices must agree exactly
    got = ora.yolact_detect(conf, boxes, msk)
    ref = t_detect(torch.from_numpy(conf), torch.from_numpy(boxes), torch.from_numpy(msk))
    assert len(got["score"]) == len(ref["score"]) > 20
    assert np.array_equal(got["prior"], ref["prior"].numpy()) and np.array_equal(got["cls"], ref["cls"].numpy())
    assert np.array_equal(got["score"], ref["score"].numpy()) and np.array_equal(got["box"], ref["box"].numpy())
    assert np.array_equal(got["mask"], ref["mask"].numpy())
    # and end to end on torch's own softmax / decode: same detections (scores within tolerance)
    ref2 = t_detect(tconf, tboxes, torch.from_numpy(msk))
    assert np.array_equal(got["prior"], ref2["prior"].numpy()) and np.array_equal(got["cls"], ref2["cls"].numpy())
    assert np.abs(got["score"] - ref2["score"].numpy()).max() <= TOL


def test_yolact_fast_nms_iou_exactly_at_threshold():
    """Boxes with IoU == 0.5 exactly survive (`iou_max <= thr`), IoU just above does not; equal scores keep index order."""
    P = 6
    boxes = np.array([[0.0, 0.0, 0.5, 0.5], [0.0, 0.0, 0.5, 0.25],        # IoU 0.5 exactly -> both kept
                      [0.5, 0.5, 1.0, 1.0], [0.5, 0.5, 1.0, 0.765625],    # IoU 0.53125 -> second suppressed
                      [0.0, 0.5, 0.25, 0.75], [0.0, 0.5, 0.25, 0.75]], np.float32)  # duplicates, equal scores
    conf = np.full((P, 81), 1e-4, np.float32)
    conf[:, 3] = [0.9, 0.8, 0.7, 0.6, 0.5, 0.5]
    msk = np.zeros((P, 32), np.float32)
    got = ora.yolact_detect(conf, boxes, msk)
    ref = t_detect(torch.from_numpy(conf), torch.from_numpy(boxes), torch.from_numpy(msk))
    assert np.array_equal(got["prior"], ref["prior"].numpy()) and np.array_equal(got["cls"], ref["cls"].numpy())
    assert np.array_equal(got["score"], ref["score"].numpy())
    # the class that carries the planted scores (the other 79 classes run the same boxes at score 1e-4, as upstream does
    # without a second threshold, and fill the rest of the 100 slots)
    assert list(got["prior"][got["cls"] == 2]) == [0, 1, 2, 4]


@pytest.mark.parametrize("hw", [(138, 138, 550, 550), (50, 50, 97, 203)])
def test_yolact_sanitize_crop_upsample_against_torch(hw):
    PH, PW, h, w = hw
    rng = np.random.default_rng(5)
    n = 12
    proto = np.maximum(rng.standard_normal((PH, PW, 32)), 0).astype(np.float32)
    coeff = np.tanh(rng.standard_normal((n, 32))).astype(np.float32)
    c = rng.uniform(0.1, 0.9, (n, 2)); s = rng.uniform(0.02, 0.6, (n, 2))
    boxes = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    boxes[0] = [0.7, 0.6, 0.2, 0.1]       # x1 > x2, y1 > y2: sanitize swaps
    boxes[1] = [-0.2, -0.1, 1.3, 1.2]     # beyond the image: clamped
    boxes[2] = [0.5, 0.5, 0.5, 0.5]       # empty box (only the 1-px padding remains)
    lo = ora.yolact_proto_masks(proto, coeff, boxes)
    masks, ib = ora.yolact_masks(proto, coeff, boxes, h, w)
    tlo, tup, thard, tb = t_postprocess(torch.from_numpy(proto), torch.from_numpy(coeff), torch.from_numpy(boxes), w, h)
    assert np.array_equal(ib, tb.numpy())                       # integer boxes: exact
    assert np.abs(lo - tlo.numpy()).max() <= 2e-6               # sigmoid(proto @ coeff) * crop window (BLAS sum order differs)
    assert np.array_equal(lo == 0, tlo.numpy() == 0)            # the crop window itself: exact
    # thresholded masks: identical except where the interpolated value sits within 1e-5 of 0.5 in the torch restatement
    diff = masks != thard.numpy().astype(np.uint8)
    assert not (diff & (np.abs(tup.numpy() - 0.5) > 1e-5)).any()
    assert diff.mean() <= 1e-5 and masks.any()


# --------------------------------------------------------------------------------- Mask R-CNN (maskrcnn-benchmark shape)
def t_boxcoder_decode(rel_codes, boxes, weights, clip=math.log(1000.0 / 16)):
    TO_REMOVE = 1
    widths = boxes[:, 2] - boxes[:, 0] + TO_REMOVE
    heights = boxes[:, 3] - boxes[:, 1] + TO_REMOVE
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = rel_codes[:, 0::4] / wx
    dy = rel_codes[:, 1::4] / wy
    dw = rel_codes[:, 2::4] / ww
    dh = rel_codes[:, 3::4] / wh
    dw = torch.clamp(dw, max=clip)
    dh = torch.clamp(dh, max=clip)
    pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
    pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
    pred_w = torch.exp(dw) * widths[:, None]
    pred_h = torch.exp(dh) * heights[:, None]
    pred = torch.zeros_like(rel_codes)
    pred[:, 0::4] = pred_ctr_x - 0.5 * pred_w
    pred[:, 1::4] = pred_ctr_y - 0.5 * pred_h
    pred[:, 2::4] = pred_ctr_x + 0.5 * pred_w - 1
    pred[:, 3::4] = pred_ctr_y + 0.5 * pred_h - 1
    return pred


def t_clip(b, w, h):
    b = b.clone()
    b[:, 0::4].clamp_(min=0, max=w - 1); b[:, 1::4].clamp_(min=0, max=h - 1)
    b[:, 2::4].clamp_(min=0, max=w - 1); b[:, 3::4].clamp_(min=0, max=h - 1)
    return b


def t_nms(boxes, scores, thr, max_keep=0):
    """The greedy loop of the lineage's nms kernel: visit by descending score, drop every later box with IoU > thr (+1 areas)."""
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order]
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    n = b.size(0)
    dead = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(int(order[i]))
        if max_keep and len(keep) == max_keep:
            break
        xx1 = torch.max(b[i, 0], b[i + 1:, 0]); yy1 = torch.max(b[i, 1], b[i + 1:, 1])
        xx2 = torch.min(b[i, 2], b[i + 1:, 2]); yy2 = torch.min(b[i, 3], b[i + 1:, 3])
        w = torch.clamp(xx2 - xx1 + 1, min=0); h = torch.clamp(yy2 - yy1 + 1, min=0)
        inter = w * h
        iou = inter / (area[i] + area[i + 1:] - inter)
        dead[i + 1:] |= iou > thr
    return np.asarray(keep, np.int64)


def _boxes(rng, n, W=1333, H=800, clustered=False):
    if clustered:
        c = rng.uniform(0.3, 0.7, (n, 2)) * (W, H) + rng.standard_normal((n, 2)) * 12
    else:
        c = rng.uniform(0, 1, (n, 2)) * (W, H)
    s = np.exp(rng.uniform(np.log(16), np.log(512), (n, 2)))
    b = np.concatenate([c - s / 2, c + s / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W - 1); b[:, 1::2] = np.clip(b[:, 1::2], 0, H - 1)
    return b.astype(np.float32)


@pytest.mark.parametrize("n,clustered", [(200, False), (1000, True), (1000, False)])
def test_greedy_nms_against_torch_loop(n, clustered):
    rng = np.random.default_rng(n + clustered)
    boxes = _boxes(rng, n, clustered=clustered)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    dup = rng.integers(0, n, n // 50)           # 2 % exact score ties, some of them on identical boxes
    scores[dup] = scores[(dup + 1) % n]
    boxes[dup[: len(dup) // 2]] = boxes[(dup[: len(dup) // 2] + 1) % n]
    for thr in (0.5, 0.7):
        got = ora.nms(boxes, scores, thr)
        ref = t_nms(torch.from_numpy(boxes), torch.from_numpy(scores), thr)
        assert np.array_equal(got, ref), (n, thr)
        assert np.array_equal(ora.nms(boxes, scores, thr, max_keep=37), ref[:37])


def test_greedy_nms_iou_exactly_at_threshold():
    # +1 areas: A = 10 x 10 = 100, B = 10 x 5 = 50, inter 50, union 100 -> IoU 0.5 exactly: kept under `>`
    boxes = np.array([[0, 0, 9, 9], [0, 0, 9, 4], [20, 20, 29, 29], [20, 20, 29, 25]], np.float32)  # second pair: 60/100 = 0.6
    scores = np.array([0.9, 0.8, 0.7, 0.6], np.float32)
    got = ora.nms(boxes, scores, 0.5)
    assert np.array_equal(got, t_nms(torch.from_numpy(boxes), torch.from_numpy(scores), 0.5)) and list(got) == [0, 1, 2]


def test_rpn_level_against_torch():
    """RPNPostProcessor.forward_for_single_feature_map: sigmoid -> topk(sorted) -> decode (weights 1) -> clip -> min-size 0 -> NMS
    0.7 -> first post_nms."""
    rng = np.random.default_rng(3)
    HWA, pre, post, W, H = 3000, 600, 300, 640, 480
    logits = (rng.standard_normal(HWA) * 2).astype(np.float32)
    deltas = (rng.standard_normal((HWA, 4)) * 0.3).astype(np.float32)
    anchors = _boxes(rng, HWA, W, H, clustered=True)
    ob, os_ = ora.rpn_level(logits, deltas, anchors, pre, post, 0.7, 0.0, W, H)
    t_logits = torch.from_numpy(logits)
    obj = torch.sigmoid(t_logits)
    sc, idx = obj.topk(pre, sorted=True)
    prop = t_boxcoder_decode(torch.from_numpy(deltas)[idx], torch.from_numpy(anchors)[idx], (1.0, 1.0, 1.0, 1.0))
    prop = t_clip(prop, W, H)
    ws = prop[:, 2] - prop[:, 0] + 1; hs = prop[:, 3] - prop[:, 1] + 1
    ok = (ws >= 0.0) & (hs >= 0.0)
    prop, sc = prop[ok], sc[ok]
    # the NMS decision is made on the ORACLE's decoded boxes / scores (taken via a huge post_nms and thr 2: no suppression), so that
    # a 1-ulp exp / sigmoid difference cannot flip an IoU test; floats are compared separately
    allb, alls = ora.rpn_level(logits, deltas, anchors, pre, pre, 2.0, 0.0, W, H)
    assert len(alls) == len(sc)
    assert np.abs(allb - prop.numpy()).max() <= 2e-4 and np.abs(alls - sc.numpy()).max() <= TOL   # boxes up to 640 px: 1e-6 relative
    keep = t_nms(torch.from_numpy(allb), torch.from_numpy(alls), 0.7, max_keep=post)
    assert np.array_equal(ob, allb[keep]) and np.array_equal(os_, alls[keep]) and 50 < len(keep) <= post


@pytest.mark.parametrize("seed", [0, 1])
def test_box_postprocess_against_torch(seed):
    """PostProcessor.forward + filter_results: softmax, decode (10, 10, 5, 5), clip, per class score > 0.05 -> NMS 0.5, concat in
    class order, kthvalue cut to 100."""
    rng = np.random.default_rng(seed)
    R, ncls, W, H = 400, 81, 800, 600
    logits = rng.standard_normal((R, ncls)).astype(np.float32)
    logits[:, 0] += 1.0
    logits[rng.integers(0, R, 300), rng.integers(1, 6, 300)] += 5.0   # crowded classes 1..5 -> far more than 100 survivors
    regr = (rng.standard_normal((R, ncls * 4)) * 0.5).astype(np.float32)
    props = _boxes(rng, R, W, H, clustered=True)
    ob, os_, ol = ora.box_postprocess(logits, regr, props, W, H, cap=128)
    prob = F.softmax(torch.from_numpy(logits), -1)
    assert np.abs(ora.softmax(logits) - prob.numpy()).max() <= TOL
    boxes = t_clip(t_boxcoder_decode(torch.from_numpy(regr), torch.from_numpy(props), (10.0, 10.0, 5.0, 5.0)), W, H).reshape(R, ncls, 4)
    # selection on the oracle's own probabilities (bit-identical inputs), torch's boxes
    prob = torch.from_numpy(ora.softmax(logits))
    res_b, res_s, res_l = [], [], []
    for j in range(1, ncls):
        inds = torch.nonzero(prob[:, j] > 0.05).squeeze(1)
        if inds.numel() == 0:
            continue
        sj, bj = prob[inds, j], boxes[inds, j]
        keep = t_nms(bj, sj, 0.5)
        res_b.append(bj[keep]); res_s.append(sj[keep]); res_l.append(torch.full((len(keep),), j, dtype=torch.int64))
    rb, rs, rl = torch.cat(res_b), torch.cat(res_s), torch.cat(res_l)
    n = rs.numel()
    assert n > 100
    image_thresh, _ = torch.kthvalue(rs, n - 100 + 1)
    keep = torch.nonzero(rs >= image_thresh.item()).squeeze(1)
    rb, rs, rl = rb[keep], rs[keep], rl[keep]
    assert len(os_) == len(rs) and np.array_equal(ol, rl.numpy().astype(np.int32)) and np.array_equal(os_, rs.numpy())
    assert np.abs(ob - rb.numpy()).max() <= 2e-4


def test_masker_paste_against_torch():
    """Masker: expand_masks (pad 1, scale), expand_boxes, int32 truncation, bilinear resize (align_corners=False), > 0.5, paste."""
    rng = np.random.default_rng(8)
    n, M, H, W = 6, 28, 120, 160
    masks = rng.uniform(0, 1, (n, M, M)).astype(np.float32)
    boxes = np.array([[10.3, 12.8, 90.2, 70.9], [-5.0, -3.0, 30.0, 40.0], [100.0, 60.0, 170.0, 130.0], [50.0, 50.0, 50.4, 50.4],
                      [0.0, 0.0, 159.0, 119.0], [33.3, 44.4, 77.7, 88.8]], np.float32)
    got = ora.paste_masks(masks, boxes, H, W)
    pad = 1
    scale = float(M + 2 * pad) / M
    for i in range(n):
        pm = torch.zeros((M + 2 * pad, M + 2 * pad)); pm[pad:-pad, pad:-pad] = torch.from_numpy(masks[i])
        b = torch.from_numpy(boxes[i])
        w_half = (b[2] - b[0]) * 0.5 * scale; h_half = (b[3] - b[1]) * 0.5 * scale
        xc = (b[2] + b[0]) * 0.5; yc = (b[3] + b[1]) * 0.5
        eb = torch.stack([xc - w_half, yc - h_half, xc + w_half, yc + h_half]).to(torch.int32)
        w = max(int(eb[2] - eb[0] + 1), 1); h = max(int(eb[3] - eb[1] + 1), 1)
        up = F.interpolate(pm[None, None], size=(h, w), mode="bilinear", align_corners=False)[0, 0]
        hard = (up > 0.5).to(torch.uint8)
        im = torch.zeros((H, W), dtype=torch.uint8)
        x0, x1 = max(int(eb[0]), 0), min(int(eb[2]) + 1, W)
        y0, y1 = max(int(eb[1]), 0), min(int(eb[3]) + 1, H)
        if x1 > x0 and y1 > y0:
            im[y0:y1, x0:x1] = hard[(y0 - int(eb[1])):(y1 - int(eb[1])), (x0 - int(eb[0])):(x1 - int(eb[0]))]
        d = got[i] != im.numpy()
        # identical except where torch's interpolated value sits within 1e-5 of the threshold
        near = torch.zeros((H, W), dtype=torch.bool)
        if x1 > x0 and y1 > y0:
            near[y0:y1, x0:x1] = (up - 0.5).abs()[(y0 - int(eb[1])):(y1 - int(eb[1])), (x0 - int(eb[0])):(x1 - int(eb[0]))] <= 1e-5
        assert not (d & ~near.numpy()).any(), i
    assert got.any()


@pytest.mark.parametrize("hw,size", [((123, 171), 200), ((480, 640), 550), ((600, 400), 64), ((550, 550), 550)])
def test_front_end_against_torch(hw, size):
    """Y1 FastBaseTransform = F.interpolate(bilinear, align_corners=False) + (x - mean) / std + BGR -> RGB, M1 = x - PIXEL_MEAN + zero padding,
    against torch on the CPU (a second opinion on the restatement, not the reference).  Tolerance: the source coordinate (dst + .5) * scale - .5
    is rounded at the magnitude of the image size, so the interpolation weight carries ~1 ulp(size) of noise whatever the association --
    times the 0..255 range over std (57): 3 ulp(max(h, w)) * 255 / 57"""
    rng = np.random.default_rng(hw[0] + size)
    x = rng.integers(0, 256, (2,) + hw + (3,), dtype=np.uint8)
    t = F.interpolate(torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2), (size, size), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    t = ((t - torch.tensor(ora.YOLACT_MEANS)) / torch.tensor(ora.YOLACT_STD)).flip(-1).numpy()
    got = ora.fast_base_transform(x, size)
    assert got.shape == t.shape and np.max(np.abs(got - t)) < 3 * float(np.spacing(np.float32(max(hw)))) * 255 / 57
    ims = [x[0], x[1, : hw[0] - 5, : hw[1] - 9]]
    out, sizes = ora.to_image_list(ims)
    H, W = -(-hw[0] // 32) * 32, -(-hw[1] // 32) * 32
    want = torch.zeros(2, H, W, 3)
    for i, im in enumerate(ims):
        want[i, : im.shape[0], : im.shape[1]] = torch.from_numpy(im.astype(np.float32)) - torch.tensor(ora.PIXEL_MEAN)
    assert np.array_equal(out, want.numpy()) and sizes.tolist() == [list(im.shape[:2]) for im in ims]
