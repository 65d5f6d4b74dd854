"""COCODemo / inference() / Yolact eval-style output on the GPU: API shape of README.md:288-335 and 243-249,
results consistent with the oracle pipeline run on the same transformed input."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cocodemo_run_and_inference(ffi, tmp_path):
    from isegmi.coco import rle_decode
    from isegmi.maskrcnn import prepare_images
    from isegmi.predictor import COCODemo, inference
    from isegmi.transforms import maskrcnn_resize
    from isegmi.weights import maskrcnn_state_dict
    from oracle.maskrcnn_ref import MaskRCNNRef
    sd = maskrcnn_state_dict(1234)
    rng = np.random.default_rng(4)
    image = rng.integers(0, 256, (150, 200, 3)).astype(np.uint8)      # HxWx3 uint8 BGR, as README.md:327-328 builds it
    demo = COCODemo(None, min_image_size=192, confidence_threshold=0.2, state_dict=sd, max_image_size=320)
    pred = demo.compute_prediction(image)
    assert pred.size == (200, 150) and pred.get_field("mask").shape[1:] == (1, 150, 200)
    # oracle on the same transformed input
    x, hw = prepare_images([maskrcnn_resize(image, 192, 320)])
    rd = MaskRCNNRef(sd).forward(x, hw)[0]
    ratio = (np.float32(200 / hw[0, 1]), np.float32(150 / hw[0, 0]))
    rm, rb = MaskRCNNRef.paste(rd, 150, 200, ratio)
    assert np.array_equal(pred.bbox, rb) and np.array_equal(pred.get_field("scores"), rd["score"])
    assert np.array_equal(pred.get_field("mask")[:, 0], rm)
    top = demo.select_top_predictions(pred)
    s = top.get_field("scores")
    assert np.all(s > 0.2) and np.all(np.diff(s) <= 0)
    out = demo.run_on_opencv_image(image)
    assert out.shape == image.shape and out.dtype == np.uint8 and (len(top) == 0 or np.any(out != image))
    res = inference(demo, [image], image_ids=[17])
    assert len(res) == len(pred)
    (tmp_path / "r.json").write_text(json.dumps(res))
    k = int(np.argmax([r["score"] for r in res]))
    assert res[k]["image_id"] == 17 and np.array_equal(rle_decode(res[k]["segmentation"]), pred.get_field("mask")[k, 0])
    demo.close()


def test_cocodemo_with_the_c4_yaml_of_the_readme(ffi):
    """README.md:313-331 with the yaml README.md:263-284 prints (R-50-C4): cfg.merge_from_file -> COCODemo -> run_on_opencv_image."""
    import os
    from isegmi.config import cfg
    from isegmi.predictor import COCODemo
    c = cfg.clone()
    c.merge_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs", "e2e_mask_rcnn_R_50_C4_1x.yaml"))
    c.MODEL.WEIGHT = "random"
    demo = COCODemo(c, min_image_size=160, confidence_threshold=0.06, max_image_size=256)
    assert demo.cfg.is_c4
    image = np.random.default_rng(8).integers(0, 256, (120, 180, 3)).astype(np.uint8)
    pred = demo.compute_prediction(image)
    assert pred.size == (180, 120) and pred.get_field("mask").shape[1:] == (1, 120, 180)
    out = demo.run_on_opencv_image(image)
    assert out.shape == image.shape and out.dtype == np.uint8


def test_yolact_eval_style_output(ffi):
    from isegmi.coco import rle_decode, yolact_results
    from isegmi.transforms import yolact_transform
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, postprocess
    rng = np.random.default_rng(8)
    frame = rng.integers(0, 256, (120, 160, 3)).astype(np.uint8)
    net = Yolact(yolact_state_dict(1234), max_batch=1, input_size=200)
    preds = net(yolact_transform(frame, 200))
    classes, scores, boxes, masks = postprocess(preds, 160, 120, score_threshold=0.15)   # --score_threshold=0.15 (README.md:243)
    top_k = 15                                                                          # --top_k=15
    classes, scores, boxes, masks = classes[:top_k], scores[:top_k], boxes[:top_k], masks[:top_k]
    assert masks.shape[1:] == (120, 160) and boxes.dtype == np.int64 and np.all(scores > 0.15) and np.all(np.diff(scores) <= 0)
    res = yolact_results(3, classes, scores, boxes, masks)
    assert len(res) == len(scores)
    if res:
        assert np.array_equal(rle_decode(res[0]["segmentation"]), masks[0])
    net.close()


def test_cli_eval_and_test_net(ffi, tmp_path):
    """`eval` (README.md:243-249) and `test_net` (README.md:344-347) front ends on a folder of synthetic PNGs."""
    from PIL import Image
    from isegmi import cli
    rng = np.random.default_rng(3)
    src = tmp_path / "in"; dst = tmp_path / "out"; src.mkdir()
    for i in range(2):
        Image.fromarray(rng.integers(0, 256, (90 + 10 * i, 120, 3)).astype(np.uint8)).save(src / ("im%d.png" % i))
    res = cli.main(["eval", "--trained_model=random", "--score_threshold=0.15", "--top_k=15", "--images=%s:%s" % (src, dst),
                    "--output_coco_json=%s" % (tmp_path / "y.json")])
    assert len(list(dst.iterdir())) == 2 and json.load(open(tmp_path / "y.json")) == res
    assert all(set(r) >= {"image_id", "category_id", "bbox", "score", "segmentation"} for r in res)
    cfgp = tmp_path / "c.yaml"
    cfgp.write_text('MODEL:\n  WEIGHT: "random"\n  ROI_HEADS:\n    DETECTIONS_PER_IMG: 20\nINPUT:\n  MIN_SIZE_TEST: 160\n  MAX_SIZE_TEST: 256\n')
    out = cli.main(["test_net", "--config-file", str(cfgp), "--images", str(src), "--output", str(tmp_path / "m.json")])
    back = json.load(open(tmp_path / "m.json"))
    assert len(back) == len(out) and {r["image_id"] for r in back} <= {0, 1} and all(len([r for r in back if r["image_id"] == i]) <= 20 for i in (0, 1))
