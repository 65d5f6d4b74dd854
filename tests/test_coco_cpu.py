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
