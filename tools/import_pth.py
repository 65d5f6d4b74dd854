"""Pretrained-weight importer: the checkpoints the reference README points at -> an .npz state dict in the
upstream key names isegmi.yolact.Yolact / isegmi.maskrcnn.MaskRCNN consume.

    python tools/import_pth.py IN OUT.npz [--family maskrcnn_r50_fpn|maskrcnn_r101_fpn|maskrcnn_r50_c4|yolact_resnet50|yolact_base|
                                                    yolact_im700|yolact_darknet53|yolact_plus_resnet50|yolact_plus_base]

Accepted inputs (torch is used HERE only to unpickle -- a tool, not the product):
  * dbolya/yolact `.pth` (the tables at README.md:209-221): a flat state dict; `module.` prefixes, `num_batches_tracked`, the
    pre-rename `backbone.layer*` duplicates and surplus `fpn.downsample_layers.*` are dropped as upstream's `load_weights` does;
  * maskrcnn-benchmark `.pth` (README.md:317 `MODEL.WEIGHT`): `{"model": {...}, "optimizer": ..., "scheduler": ..., "iteration": n}`
    with `module.` prefixes;
  * Detectron / Caffe2 `.pkl` (README.md:266 `catalog://ImageNetPretrained/MSRA/R-50` and the Detectron model zoo): `{"blobs": {...}}`
    or a flat dict of numpy blobs named `conv1_w`, `res2_0_branch2a_w`, `res2_0_branch2a_bn_s`, `fpn_inner_res5_2_sum_w`,
    `conv_rpn_fpn2_w`, `fc6_w`, `_[mask]_fcn1_w`, `conv5_mask_w` ...: renamed with the rules of maskrcnn-benchmark's
    utils/c2_model_loading.py and attached to the full module path by longest-suffix match against the family's key set; the
    AffineChannel scale / bias become FrozenBatchNorm weight / bias with running_mean 0 and running_var 1 (what an untouched
    FrozenBatchNorm2d buffer holds upstream); momentum blobs and `fpn3..6` copies of the shared RPN head are dropped.

With --family the result is checked against the family's complete key set (isegmi.weights generators): missing or unused keys
are an error unless --allow-partial (backbone-only ImageNet checkpoints).
"""
import argparse
import os
import pickle
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "instancesegmentation-jittor_amd")]

FAMILIES = ("maskrcnn_r50_fpn", "maskrcnn_r101_fpn", "maskrcnn_r50_c4", "yolact_resnet50", "yolact_base", "yolact_im700",
            "yolact_darknet53", "yolact_plus_resnet50", "yolact_plus_base")


def expected_keys(family):
    """The complete state-dict key set (and shapes) of a model family, from the synthetic generators."""
    from isegmi import weights as W
    from isegmi.yolact import YolactConfig
    if family == "maskrcnn_r50_fpn":
        sd = W.maskrcnn_state_dict(0, 50)
    elif family == "maskrcnn_r101_fpn":
        sd = W.maskrcnn_state_dict(0, 101)
    elif family == "maskrcnn_r50_c4":
        sd = W.maskrcnn_c4_state_dict(0)
    else:
        cfg = {"yolact_resnet50": YolactConfig(), "yolact_base": YolactConfig.base(), "yolact_im700": YolactConfig.im700(),
               "yolact_darknet53": YolactConfig.darknet53(), "yolact_plus_resnet50": YolactConfig.plus_resnet50(),
               "yolact_plus_base": YolactConfig.plus_base()}[family]
        sd = W.yolact_state_dict(0, cfg.depth, cfg.num_priors, cfg.dcn_layers, cfg.dcn_interval, cfg.use_maskiou, cfg.backbone)
    return {k: v.shape for k, v in sd.items()}


def _to_numpy(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def load_any(path):
    """.pth / .pt through torch.load, .pkl through pickle (latin1: Caffe2 pickles are Python 2)."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            return pickle.load(f, encoding="latin1")
    import torch
    return torch.load(path, map_location="cpu", weights_only=False)


def is_caffe2(keys):
    return any(k in keys for k in ("conv1_w", "res_conv1_bn_s")) or sum(k.endswith("_w") for k in keys) > len(keys) // 4


def rename_caffe2(k):
    """One Detectron blob name -> maskrcnn-benchmark module-relative name (utils/c2_model_loading.py, restated)."""
    k = k.replace("_", ".").replace(".w", ".weight").replace(".bn", "_bn").replace(".b", ".bias")
    k = k.replace("_bn.s", "_bn.scale").replace(".biasranch", ".branch").replace("bbox.pred", "bbox_pred").replace("cls.score", "cls_score")
    k = k.replace("res.conv1_", "conv1_").replace(".biasbox", ".bbox").replace("conv.rpn", "rpn.conv")
    k = k.replace("rpn.bbox.pred", "rpn.bbox_pred").replace("rpn.cls.logits", "rpn.cls_logits")
    k = k.replace("_bn.scale", "_bn.weight").replace("conv1_bn.", "bn1.")
    for c2, mb in (("res2.", "layer1."), ("res3.", "layer2."), ("res4.", "layer3."), ("res5.", "layer4.")):
        k = k.replace(c2, mb)
    for c2, mb in ((".branch2a.", ".conv1."), (".branch2a_bn.", ".bn1."), (".branch2b.", ".conv2."), (".branch2b_bn.", ".bn2."),
                   (".branch2c.", ".conv3."), (".branch2c_bn.", ".bn3."), (".branch1.", ".downsample.0."), (".branch1_bn.", ".downsample.1.")):
        k = k.replace(c2, mb)
    # FPN: fpn.inner.layerN.M.sum(.lateral) -> fpn_innerN; fpn.layerN.M.sum -> fpn_layerN (the stage's LAST block index M is dropped)
    m = re.match(r"fpn\.inner\.layer(\d)\.\d+\.sum(?:\.lateral)?\.(weight|bias)$", k)
    if m:
        return "fpn_inner%s.%s" % (m.group(1), m.group(2))
    m = re.match(r"fpn\.layer(\d)\.\d+\.sum\.(weight|bias)$", k)
    if m:
        return "fpn_layer%s.%s" % (m.group(1), m.group(2))
    k = k.replace("rpn.conv.fpn2", "rpn.conv").replace("rpn.bbox_pred.fpn2", "rpn.bbox_pred").replace("rpn.cls_logits.fpn2", "rpn.cls_logits")
    # mask head
    k = k.replace("mask.fcn.logits", "mask_fcn_logits").replace(".[mask].fcn", "mask_fcn").replace("conv5.mask", "conv5_mask")
    if k.startswith("rpn."):   # RPNModule.head
        k = "rpn.head." + k[4:]
    return k


def convert(sd, family=None, allow_partial=False):
    """Any of the accepted checkpoint dicts -> {upstream key: float32 ndarray}.  With `family`, keys are attached to the family's
    full module paths (Caffe2 input) and the result is validated against its complete key set."""
    for wrap in ("model", "state_dict", "blobs"):
        if isinstance(sd, dict) and wrap in sd and isinstance(sd[wrap], dict):
            sd = sd[wrap]
    flat = {}
    for k, v in sd.items():
        if not hasattr(v, "shape") or k.endswith("num_batches_tracked") or k.endswith("_momentum"):
            continue
        flat[k[7:] if k.startswith("module.") else k] = _to_numpy(v)
    out = {}
    if is_caffe2(set(flat)):
        if family is None or not family.startswith("maskrcnn"):
            raise ValueError("a Detectron / Caffe2 checkpoint needs --family maskrcnn_*")
        exp = expected_keys(family)
        short = {}
        for k, v in flat.items():
            if re.search(r"_fpn[3-6]_", k) or k in ("lr", "weight_order"):   # the RPN head is one shared module: fpn2 carries it
                continue
            short[rename_caffe2(k)] = (k, v)
        used = set()
        for full in exp:   # maskrcnn-benchmark's align_and_update_state_dicts: the LONGEST loaded name that is a suffix of the module path
            cands = [s_ for s_ in short if full == s_ or full.endswith("." + s_)]
            if not cands:
                continue
            best = max(cands, key=len)
            used.add(best)
            out[full] = np.asarray(short[best][1], np.float32).reshape(exp[full])
        stray = sorted(short[s_][0] for s_ in set(short) - used)
        if stray and not allow_partial:
            raise KeyError("%d blobs match no module of %s: %s ..." % (len(stray), family, ", ".join(stray[:5])))
        for k in list(out):  # AffineChannel -> FrozenBatchNorm: identity running statistics
            if re.search(r"(bn\d|downsample\.1)\.weight$", k):
                p = k[: -len(".weight")]
                out.setdefault(p + ".running_mean", np.zeros_like(out[k]))
                out.setdefault(p + ".running_var", np.ones_like(out[k]))
    else:
        for k, v in flat.items():
            if k.startswith("backbone.layer") and not k.startswith("backbone.layers"):   # yolact load_weights: pre-rename copies
                continue
            m = re.match(r"fpn\.downsample_layers\.(\d+)\.", k)
            if m and int(m.group(1)) >= 2:
                continue
            out[k] = np.asarray(v, np.float32)
    if family is not None:
        exp = expected_keys(family)
        missing = sorted(set(exp) - set(out))
        unused = sorted(set(out) - set(exp))
        bad = sorted(k for k in out if k in exp and tuple(out[k].shape) != tuple(exp[k]))
        if bad:
            raise ValueError("shape mismatch for %s: %s" % (family, ", ".join("%s %s != %s" % (k, out[k].shape, exp[k]) for k in bad[:5])))
        if unused or (missing and not allow_partial):
            raise KeyError("%s: %d missing (%s ...), %d unused (%s ...)" % (family, len(missing), ", ".join(missing[:4]), len(unused), ", ".join(unused[:4])))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--family", choices=FAMILIES, default=None)
    ap.add_argument("--allow-partial", action="store_true", help="backbone-only checkpoints (ImageNet-pretrained R-50 / R-101)")
    a = ap.parse_args()
    out = convert(load_any(a.src), a.family, a.allow_partial)
    np.savez(a.dst, **out)
    print("wrote %d tensors, %.1f M parameters" % (len(out), sum(v.size for v in out.values()) / 1e6))


if __name__ == "__main__":
    main()
