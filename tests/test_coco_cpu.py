"""SURVEY 8(f) rows on the CPU: RLE encoder (pycocotools algorithm), COCO result records, resize rule, importer."""
import json
import os

import numpy as np
import pytest


def test_rle_known_answers_and_roundtrip():
    from isegmi.coco import rle_counts, rle_decode, rle_encode, rle_from_string, rle_to_string
    m = np.array([[0, 1], [1, 1]], np.uint8)             # column-major: 0 1 1 1
    assert rle_counts(m) == [1, 3] and rle_to_string([1, 3]) == "13"
    assert rle_counts(np.ones((2, 2), np.uint8)) == [0, 4]
    assert rle_counts(np.zeros((3, 2), np.uint8)) == [6]
    # multi-character counts and the delta-vs-two-back rule (i > 2), incl. negative deltas
    counts = [5, 100, 2000, 40, 1, 70000, 3]
    s = rle_to_string(counts)
    assert rle_from_string(s) == counts and all(48 <= ord(c) < 48 + 64 for c in s)
    assert rle_to_string([32]) == "P1"                     # 32: low 5 bits 0 + continuation bit, then 1
    rng = np.random.default_rng(0)
    for shape in ((17, 23), (1, 50), (64, 1), (120, 160)):
        mm = (rng.uniform(0, 1, shape) < 0.3).astype(np.uint8)
        mm[5 % shape[0]:, : shape[1] // 2] = 1
        r = rle_encode(mm)
        assert r["size"] == list(shape) and np.array_equal(rle_decode(r), mm)
        assert sum(rle_from_string(r["counts"])) == mm.size


def test_result_records_and_json(tmp_path):
    from isegmi.coco import COCO_CATEGORY_IDS, dump, maskrcnn_results, yolact_results
    assert len(COCO_CATEGORY_IDS) == 80 and COCO_CATEGORY_IDS[0] == 1 and COCO_CATEGORY_IDS[-1] == 90 and COCO_CATEGORY_IDS[11] == 13
    masks = np.zeros((2, 10, 12), np.uint8); masks[0, 2:5, 3:9] = 1
    r = maskrcnn_results(42, [[3, 2, 8, 4], [0, 0, 11, 9]], [0.9, 0.3], [1, 80], masks)
    assert r[0]["bbox"] == [3.0, 2.0, 6.0, 3.0] and r[0]["category_id"] == 1 and r[1]["category_id"] == 90 and r[0]["image_id"] == 42
    assert r[0]["segmentation"]["size"] == [10, 12]
    y = yolact_results(7, [0, 79], [0.5, 0.25], np.array([[3, 2, 9, 5], [0, 0, 12, 10]], np.int64), masks)
    assert y[0]["bbox"] == [3.0, 2.0, 6.0, 3.0] and y[1]["category_id"] == 90
    p = tmp_path / "res.json"
    dump(r + y, str(p))
    back = json.load(open(p))
    assert len(back) == 4 and isinstance(back[0]["segmentation"]["counts"], str)


def test_resize_rule_and_bilinear():
    torch = pytest.importorskip("torch")
    from isegmi.transforms import bilinear_resize, get_size, maskrcnn_resize
    assert get_size(640, 480) == (800, 1066)        # short side 480 -> 800, int(800*640/480)
    assert get_size(1000, 300) == (400, 1333)       # long side capped at 1333
    assert get_size(1066, 800) == (800, 1066)       # already at size
    assert get_size(480, 640) == (1066, 800)
    img = np.random.default_rng(0).integers(0, 256, (48, 64, 3)).astype(np.uint8)
    out = maskrcnn_resize(img, 96, 200)
    assert out.shape == (96, 128, 3) and out.dtype == np.float32
    x = img.astype(np.float32)
    ref = torch.nn.functional.interpolate(torch.from_numpy(x).permute(2, 0, 1)[None], size=(55, 55), mode="bilinear",
                                          align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.max(np.abs(bilinear_resize(x, 55, 55) - ref)) < 1e-3


def test_pth_importer_roundtrip():
    torch = pytest.importorskip("torch")
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("import_pth", os.path.join(root, "tools", "import_pth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from isegmi.weights import yolact_state_dict
    sd = yolact_state_dict(3)
    keys = list(sd)[:40]
    tsd = {"module." + k: torch.from_numpy(sd[k]) for k in keys}
    tsd["module.backbone.bn1.num_batches_tracked"] = torch.tensor(5)
    out = mod.convert({"model": tsd})
    assert set(out) == set(keys) and all(np.array_equal(out[k], sd[k]) for k in out)


# ---------------------------------------------------------------------------------------------------------------------------
# Complete checkpoints in the formats the reference README points at (README.md:209-221 Yolact .pth tables; README.md:317
# maskrcnn-benchmark MODEL.WEIGHT .pth; README.md:266 Detectron / Caffe2 pkl naming) through tools/import_pth.convert and
# then through the engines' own weight loaders, without a missing or unused key.
def _importer():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("import_pth", os.path.join(root, "tools", "import_pth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Tracking(dict):
    """state dict that records which keys the loader read"""

    def __init__(self, *a):
        super().__init__(*a)
        self.read = set()

    def __getitem__(self, k):
        self.read.add(k)
        return super().__getitem__(k)

    def __contains__(self, k):
        return super().__contains__(k)


def _dry_load_maskrcnn(sd, cfg, H=64, W=96):
    """MaskRCNN._load / _load_c4 with the device calls recorded instead of executed (no GPU here)."""
    from isegmi.maskrcnn import MaskRCNN
    m = object.__new__(MaskRCNN)
    m.cfg, m.H, m.W, m._h = cfg, H, W, None
    m.convs, m.tensors = {}, {}
    m._set_conv_krsc = lambda name, w, scale=None, shift=None: m.convs.__setitem__(name, np.asarray(w).shape)
    m._set_tensor = lambda name, a: m.tensors.__setitem__(name, np.asarray(a).shape)
    m.params = {}
    m.set_param = lambda name, value: m.params.__setitem__(name, value)
    t = _Tracking(sd)
    (m._load_c4 if cfg.is_c4 else m._load)(t)
    return m, t


def _dry_load_yolact(sd, cfg, size=550):
    from isegmi.yolact import Yolact
    n = object.__new__(Yolact)
    n.cfg, n.size, n._h, n.fuse_heads = cfg, size, None, True
    n.convs, n.tensors = {}, {}
    n._set_conv = lambda name, w, scale=None, shift=None, pad_cin_to=None: n.convs.__setitem__(name, np.asarray(w).shape)
    n._set_tensor = lambda name, a: n.tensors.__setitem__(name, np.asarray(a).shape)
    t = _Tracking(sd)
    n._load(t)
    return n, t


def _c2_name(k, depth):
    """maskrcnn-benchmark key -> the Detectron / Caffe2 blob name (written out from the Detectron naming scheme, independently
    of the importer's rename rules); None for tensors a Caffe2 checkpoint does not hold (FrozenBN running statistics)."""
    last = {1: 2, 2: 3, 3: 22 if depth == 101 else 5, 4: 2}
    sfx = {"weight": "w", "bias": "b"}
    p = k.split(".")
    if k.endswith("running_mean") or k.endswith("running_var"):
        return None
    if k.startswith("backbone.body.stem."):
        return "conv1_w" if p[3] == "conv1" else "res_conv1_bn_" + {"weight": "s", "bias": "b"}[p[4]]
    if k.startswith("backbone.body.layer") or ".head.layer4." in k:
        i = p.index([x for x in p if x.startswith("layer")][0])
        li, b, mod = int(p[i][5:]), p[i + 1], p[i + 2]
        br = {"conv1": "branch2a", "bn1": "branch2a_bn", "conv2": "branch2b", "bn2": "branch2b_bn", "conv3": "branch2c", "bn3": "branch2c_bn"}
        if mod == "downsample":
            return "res%d_%s_branch1%s" % (li + 1, b, "_w" if p[i + 3] == "0" else "_bn_" + {"weight": "s", "bias": "b"}[p[i + 4]])
        return "res%d_%s_%s_%s" % (li + 1, b, br[mod], sfx[p[i + 3]] if not mod.startswith("bn") else {"weight": "s", "bias": "b"}[p[i + 3]])
    if k.startswith("backbone.fpn.fpn_inner"):
        n = int(p[2][-1])
        return "fpn_inner_res%d_%d_sum%s_%s" % (n + 1, last[n], "" if n == 4 else "_lateral", sfx[p[3]])
    if k.startswith("backbone.fpn.fpn_layer"):
        n = int(p[2][-1])
        return "fpn_res%d_%d_sum_%s" % (n + 1, last[n], sfx[p[3]])
    if k.startswith("rpn.head."):
        return {"conv": "conv_rpn_fpn2", "cls_logits": "rpn_cls_logits_fpn2", "bbox_pred": "rpn_bbox_pred_fpn2"}[p[2]] + "_" + sfx[p[3]]
    if ".mask_fcn" in k and "logits" not in k:
        return "_[mask]_fcn%s_%s" % (p[3][-1], sfx[p[4]])
    return p[-2] + "_" + sfx[p[-1]]   # fc6 fc7 cls_score bbox_pred conv5_mask mask_fcn_logits


@pytest.mark.parametrize("family,depth", [("maskrcnn_r50_fpn", 50), ("maskrcnn_r101_fpn", 101)])
def test_importer_takes_complete_maskrcnn_checkpoints(tmp_path, family, depth):
    torch = pytest.importorskip("torch")
    mod = _importer()
    from isegmi.maskrcnn import MaskRCNNConfig
    from isegmi.weights import maskrcnn_state_dict
    sd = maskrcnn_state_dict(11, depth)   # random tensors under the upstream names / shapes (OIHW convs, [out, in] FCs)
    # (a) maskrcnn-benchmark .pth: {"model": {"module.<key>": tensor}, "optimizer": ..., "iteration": n}, through a real file
    ckpt = {"model": {"module." + k: torch.from_numpy(v) for k, v in sd.items()}, "optimizer": {"state": {}, "param_groups": [{"lr": 0.02}]},
            "scheduler": {"last_epoch": 90000}, "iteration": 90000}
    path = tmp_path / "model_final.pth"
    torch.save(ckpt, path)
    out = mod.convert(mod.load_any(str(path)), family)
    assert set(out) == set(sd) and all(np.array_equal(out[k], sd[k]) for k in sd)
    m, t = _dry_load_maskrcnn(out, MaskRCNNConfig(depth=depth))
    assert t.read == set(out), sorted(set(out) - t.read)[:5]          # the loader consumed every tensor of the checkpoint
    assert "backbone.body.layer3.%d.conv3" % (22 if depth == 101 else 5) in m.convs and "anchor_base.4" in m.tensors
    # (b) Detectron / Caffe2 pkl naming ({"blobs": {...}} with momentum blobs and the per-level copies of the shared RPN head)
    blobs = {}
    for k, v in sd.items():
        c2 = _c2_name(k, depth)
        if c2 is not None:
            blobs[c2] = v
            blobs[c2 + "_momentum"] = np.zeros_like(v)
    for lvl in (3, 4, 5, 6):
        for nm in ("conv_rpn_fpn%d_w", "rpn_cls_logits_fpn%d_b"):
            blobs[nm % lvl] = blobs[nm % 2]
    import pickle
    pk = tmp_path / "model_final.pkl"
    with open(pk, "wb") as f:
        pickle.dump({"blobs": blobs, "cfg": "..."}, f, protocol=2)
    out2 = mod.convert(mod.load_any(str(pk)), family)
    assert set(out2) == set(sd)
    for k in sd:
        if k.endswith("running_mean"):
            assert not out2[k].any()
        elif k.endswith("running_var"):
            assert (out2[k] == 1).all()
        else:
            assert np.array_equal(out2[k], sd[k]), k
    _, t2 = _dry_load_maskrcnn(out2, MaskRCNNConfig(depth=depth))
    assert t2.read == set(out2)
    # a stray blob and a missing one are errors, not silence
    with pytest.raises(KeyError):
        mod.convert({"blobs": dict(blobs, foo_bar_w=np.zeros(3, np.float32))}, family)
    short = dict(blobs); short.pop("fc7_w")
    with pytest.raises(KeyError):
        mod.convert({"blobs": short}, family)
    # backbone-only ImageNet checkpoint (README.md:266 catalog://ImageNetPretrained/MSRA/R-50): partial import allowed on request
    bb = {c: v for c, v in blobs.items() if c.startswith(("conv1", "res")) and not c.endswith("_momentum")}
    part = mod.convert({"blobs": bb}, family, allow_partial=True)
    assert all(k.startswith("backbone.body.") for k in part) and "backbone.body.stem.conv1.weight" in part


def test_importer_takes_complete_c4_and_yolact_checkpoints(tmp_path):
    torch = pytest.importorskip("torch")
    mod = _importer()
    from isegmi.maskrcnn import MaskRCNNConfig
    from isegmi.weights import maskrcnn_c4_state_dict, yolact_state_dict
    from isegmi.yolact import YolactConfig
    sd = maskrcnn_c4_state_dict(5)
    out = mod.convert({"model": {"module." + k: torch.from_numpy(v) for k, v in sd.items()}}, "maskrcnn_r50_c4")
    _, t = _dry_load_maskrcnn(out, MaskRCNNConfig.c4(), 64, 96)
    # SHARE_BOX_FEATURE_EXTRACTOR: the mask branch's copy of layer4 is the box head's module; the loader reads it once
    assert set(out) - t.read == {k for k in out if k.startswith("roi_heads.mask.feature_extractor.head.")}
    for family, cfg in (("yolact_resnet50", YolactConfig()), ("yolact_base", YolactConfig.base()), ("yolact_plus_resnet50", YolactConfig.plus_resnet50()),
                        ("yolact_darknet53", YolactConfig.darknet53())):
        sd = yolact_state_dict(3, cfg.depth, cfg.num_priors, cfg.dcn_layers, cfg.dcn_interval, cfg.use_maskiou, cfg.backbone)
        tsd = {"module." + k: torch.from_numpy(v) for k, v in sd.items()}
        for k in list(sd):   # what a real yolact .pth also carries: BN step counters, pre-rename backbone copies, surplus downsample layers
            if k.endswith("running_var"):
                tsd["module." + k[:-11] + "num_batches_tracked"] = torch.tensor(800000)
        tsd["module.backbone.layer1.0.conv1.weight"] = torch.zeros(1)
        tsd["module.fpn.downsample_layers.2.weight"] = torch.zeros(256, 256, 3, 3)
        path = tmp_path / (family + ".pth")
        torch.save(tsd, path)
        out = mod.convert(mod.load_any(str(path)), family)
        assert set(out) == set(sd) and all(np.array_equal(out[k], sd[k]) for k in sd)
        n, t = _dry_load_yolact(out, cfg)
        assert t.read == set(out), (family, sorted(set(out) - t.read)[:5])
        assert "prediction_layers.0.head_cat" in n.convs and "priors" in n.tensors
