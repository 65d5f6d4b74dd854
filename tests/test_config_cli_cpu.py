"""cfg node (README.md:313-317 usage) and CLI argument surface, on the CPU."""
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cfg_merge_and_mapping():
    from isegmi.config import cfg, to_maskrcnn_config
    c = cfg.clone()
    c.merge_from_file(os.path.join(ROOT, "configs", "e2e_mask_rcnn_R_50_FPN_1x.yaml"))
    assert c.MODEL.RPN.ANCHOR_STRIDE == (4, 8, 16, 32, 64) and c.DATASETS.TEST == ("coco_2014_minival",)
    c.merge_from_list(["MODEL.ROI_HEADS.DETECTIONS_PER_IMG", 50, "MODEL.BACKBONE.CONV_BODY", "R-101-FPN"])
    c.MODEL.WEIGHT = "weight/maskrcnn_r101.npz"               # attribute assignment as in README.md:317
    m = to_maskrcnn_config(c)
    assert m.depth == 101 and m.DETECTIONS_PER_IMG == 50 and m.RPN_PRE_NMS_TOP_N_TEST == 1000 and m.ANCHOR_STRIDE == (4, 8, 16, 32, 64)
    assert cfg.MODEL.BACKBONE.CONV_BODY == "R-50-FPN"        # the global default is untouched by the clone
    c.MODEL.RPN.PRE_NMS_TOP_N_TEST = 6000                    # fine for the C4 body, too many for a per-level FPN selection
    with pytest.raises(ValueError, match="1024"):
        to_maskrcnn_config(c)
    c.MODEL.RPN.PRE_NMS_TOP_N_TEST = 1000
    c.MODEL.BACKBONE.CONV_BODY = "R-152-FPN"
    with pytest.raises(ValueError, match="built bodies"):
        to_maskrcnn_config(c)


def test_c4_yaml_of_the_readme_maps_to_the_c4_engine_config():
    """README.md:263-284 prints e2e_mask_rcnn_R_50_C4_1x.yaml; its keys (incl. the training ones) load and map."""
    from isegmi.config import cfg, to_maskrcnn_config
    c = cfg.clone()
    c.merge_from_file(os.path.join(ROOT, "configs", "e2e_mask_rcnn_R_50_C4_1x.yaml"))
    assert c.MODEL.ROI_MASK_HEAD.SHARE_BOX_FEATURE_EXTRACTOR is True and c.SOLVER.STEPS == (120000, 160000)
    m = to_maskrcnn_config(c)
    assert m.is_c4 and m.RPN_PRE_NMS_TOP_N_TEST == 6000 and m.RPN_POST_NMS_TOP_N_TEST == 1000 and m.ANCHOR_STRIDE == (16,)
    assert m.SIZE_DIVISIBILITY == 16 and m.ANCHOR_SIZES == (32, 64, 128, 256, 512)


def test_cli_parses_reference_flags():
    from isegmi import cli
    with pytest.raises(SystemExit):
        cli.main(["eval", "--help"])
    import argparse
    # the exact flag spellings of README.md:243-249 / 344-347 must be accepted by the parser
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd")
    assert callable(cli.cmd_eval) and callable(cli.cmd_test_net)
