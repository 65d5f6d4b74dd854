"""`cfg` -- the yacs-shaped config node the reference's usage snippet drives (README.md:296, 313-317):

    from isegmi.config import cfg
    cfg.merge_from_file('configs/e2e_mask_rcnn_R_50_FPN_1x.yaml')
    cfg.MODEL.WEIGHT = "weight/maskrcnn_r50.npz"            # README.md:317
    coco_demo = COCODemo(cfg, min_image_size=800, confidence_threshold=0.5)

Only the inference keys of the hot path carry meaning; training/solver/dataset keys from the yaml
(README.md:263-284) are accepted and stored untouched.  yacs itself is not in the image: this is a ~60-line stand-in.
"""
import ast
import copy

import yaml

from .maskrcnn import MaskRCNNConfig


def _literal(v):
    """yacs semantics: yaml hands tuples over as strings like "(4, 8, 16)"; lists become tuples."""
    if isinstance(v, list):
        return tuple(v)
    if isinstance(v, str) and v[:1] in "([" and v[-1:] in ")]":
        try:
            return tuple(ast.literal_eval(v))
        except (ValueError, SyntaxError):
            return v
    return v


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                node = self.setdefault(k, CfgNode())
                if not isinstance(node, CfgNode):
                    node = self[k] = CfgNode()
                node._merge(v)
            else:
                self[k] = _literal(v)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, kv):
        assert len(kv) % 2 == 0
        for key, val in zip(kv[::2], kv[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, CfgNode())
            node[parts[-1]] = _literal(val)

    def clone(self):
        return copy.deepcopy(self)


def _defaults():
    c = CfgNode()
    c._merge({
        "INPUT": {"MIN_SIZE_TEST": 800, "MAX_SIZE_TEST": 1333, "PIXEL_MEAN": (102.9801, 115.9465, 122.7717), "TO_BGR255": True},
        "DATALOADER": {"SIZE_DIVISIBILITY": 32},
        "MODEL": {"META_ARCHITECTURE": "GeneralizedRCNN", "WEIGHT": "", "MASK_ON": True,
                  "BACKBONE": {"CONV_BODY": "R-50-FPN"},
                  "RPN": {"USE_FPN": True, "ANCHOR_SIZES": (32, 64, 128, 256, 512), "ANCHOR_STRIDE": (4, 8, 16, 32, 64),
                          "ASPECT_RATIOS": (0.5, 1.0, 2.0), "PRE_NMS_TOP_N_TEST": 1000, "POST_NMS_TOP_N_TEST": 1000,
                          "FPN_POST_NMS_TOP_N_TEST": 1000, "NMS_THRESH": 0.7, "MIN_SIZE": 0},
                  "ROI_HEADS": {"SCORE_THRESH": 0.05, "NMS": 0.5, "DETECTIONS_PER_IMG": 100},
                  "ROI_MASK_HEAD": {"PREDICTOR": "MaskRCNNC4Predictor", "RESOLUTION": 28}},
    })
    return c


cfg = _defaults()


def to_maskrcnn_config(c):
    """Map the yaml-keyed node onto the frozen dataclass the engine consumes; rejects what the path does not build."""
    body = c.MODEL.BACKBONE.CONV_BODY
    r = c.MODEL.RPN
    h = c.MODEL.ROI_HEADS
    if body == "R-50-C4":
        # the yaml the reference prints (README.md:263-273): everything it does not set comes from maskrcnn-benchmark's
        # defaults.py -- one stride-16 map, no FPN merge, SIZE_DIVISIBILITY 0 (the engine pads to 16), 14x14 masks
        if int(r.PRE_NMS_TOP_N_TEST) > 6144 or int(r.POST_NMS_TOP_N_TEST) > 1024:
            raise ValueError("MODEL.RPN.PRE/POST_NMS_TOP_N_TEST: the single-map HIP selection kernels hold 6144 / 1024 boxes")
        return MaskRCNNConfig(CONV_BODY=body, MIN_SIZE_TEST=int(c.INPUT.MIN_SIZE_TEST), MAX_SIZE_TEST=int(c.INPUT.MAX_SIZE_TEST),
                              SIZE_DIVISIBILITY=16, ANCHOR_SIZES=tuple(r.ANCHOR_SIZES), ANCHOR_STRIDE=(16,),
                              ASPECT_RATIOS=tuple(float(x) for x in r.ASPECT_RATIOS), RPN_PRE_NMS_TOP_N_TEST=int(r.PRE_NMS_TOP_N_TEST),
                              RPN_POST_NMS_TOP_N_TEST=int(r.POST_NMS_TOP_N_TEST), RPN_NMS_THRESH=float(r.NMS_THRESH),
                              RPN_MIN_SIZE=float(r.MIN_SIZE), ROI_SCORE_THRESH=float(h.SCORE_THRESH), ROI_NMS=float(h.NMS),
                              DETECTIONS_PER_IMG=int(h.DETECTIONS_PER_IMG))
    if body not in ("R-50-FPN", "R-101-FPN"):
        raise ValueError("built bodies: R-50-FPN, R-101-FPN, R-50-C4 (got %r)" % body)
    for k in ("PRE_NMS_TOP_N_TEST", "POST_NMS_TOP_N_TEST", "FPN_POST_NMS_TOP_N_TEST"):
        if int(c.MODEL.RPN[k]) > 1024:
            raise ValueError("MODEL.RPN.%s=%d: the per-level FPN selection kernels hold at most 1024 boxes" % (k, c.MODEL.RPN[k]))
    return MaskRCNNConfig(depth=101 if "101" in body else 50, CONV_BODY=body, MIN_SIZE_TEST=int(c.INPUT.MIN_SIZE_TEST),
                          MAX_SIZE_TEST=int(c.INPUT.MAX_SIZE_TEST),
                          SIZE_DIVISIBILITY=int(c.DATALOADER.SIZE_DIVISIBILITY), ANCHOR_SIZES=tuple(r.ANCHOR_SIZES),
                          ANCHOR_STRIDE=tuple(r.ANCHOR_STRIDE), ASPECT_RATIOS=tuple(float(x) for x in r.ASPECT_RATIOS),
                          RPN_PRE_NMS_TOP_N_TEST=int(r.PRE_NMS_TOP_N_TEST), RPN_POST_NMS_TOP_N_TEST=int(r.POST_NMS_TOP_N_TEST),
                          RPN_FPN_POST_NMS_TOP_N_TEST=int(r.FPN_POST_NMS_TOP_N_TEST), RPN_NMS_THRESH=float(r.NMS_THRESH),
                          RPN_MIN_SIZE=float(r.MIN_SIZE), ROI_SCORE_THRESH=float(h.SCORE_THRESH), ROI_NMS=float(h.NMS),
                          DETECTIONS_PER_IMG=int(h.DETECTIONS_PER_IMG))
