"""COCODemo -- the predictor object of README.md:288-335, backed by libisegmi.

    coco_demo = COCODemo(cfg, min_image_size=800, confidence_threshold=0.5, state_dict=sd)
    predictions = coco_demo.run_on_opencv_image(image)     # image: HxWx3 uint8 BGR -> annotated HxWx3 uint8

`cfg` is a MaskRCNNConfig (the handful of inference constants the path needs, keyed like the yaml).
compute_prediction / select_top_predictions follow the upstream demo ([UPSTREAM-RECALL]);
overlay drawing is plain numpy (cv2 is not in the image): mask tint + box outline, no text.
"""
import collections

import numpy as np

from .coco import COCO_CLASSES
from .maskrcnn import BoxList, MaskRCNN, MaskRCNNConfig
from .transforms import maskrcnn_resize_u8


class COCODemo:
    CATEGORIES = ("__background",) + COCO_CLASSES

    MAX_ENGINES = 4   # one engine per distinct padded input size (weights + activation buffers each): least recently used is closed

    def __init__(self, cfg=None, min_image_size=800, confidence_threshold=0.5, state_dict=None, max_image_size=1333, device=0):
        if cfg is not None and not isinstance(cfg, MaskRCNNConfig):   # the yacs-shaped node of isegmi.config (README.md:313-324)
            from .config import to_maskrcnn_config
            if state_dict is None and getattr(cfg.MODEL, "WEIGHT", ""):
                w = cfg.MODEL.WEIGHT
                from .weights import maskrcnn_c4_state_dict, maskrcnn_state_dict
                body = cfg.MODEL.BACKBONE.CONV_BODY
                rnd = (lambda: maskrcnn_c4_state_dict(1234)) if body.endswith("-C4") else (lambda: maskrcnn_state_dict(1234, 101 if "101" in body else 50))
                state_dict = rnd() if w == "random" else dict(np.load(w))
            cfg = to_maskrcnn_config(cfg)
        self.cfg = cfg or MaskRCNNConfig()
        if state_dict is None:
            raise ValueError("COCODemo needs weights: pass state_dict=... or set cfg.MODEL.WEIGHT to an .npz (or 'random'); "
                             "the reference's download URLs (README.md:266) cannot be fetched here")
        self.min_image_size, self.max_image_size = min_image_size, max_image_size
        self.confidence_threshold = confidence_threshold
        self.state_dict, self.device = state_dict, device
        self._models = collections.OrderedDict()

    def _model(self, H, W):
        """Engine for the padded size (H, W).  Upstream pads a single image to the next multiple of 32 only, and the RPN sees the
        padded area, so results depend on the exact canvas: one engine per canvas size, at most MAX_ENGINES alive (LRU)."""
        key = (H, W)
        if key in self._models:
            self._models.move_to_end(key)
            return self._models[key]
        while len(self._models) >= self.MAX_ENGINES:
            _, old = self._models.popitem(last=False)
            old.close()
        self._models[key] = MaskRCNN(self.state_dict, H, W, cfg=self.cfg, max_batch=1, device=self.device)
        return self._models[key]

    def compute_prediction(self, original_image):
        """-> BoxList in ORIGINAL image coordinates with scores, labels and mask [n,1,H,W] uint8 (Masker output)."""
        h, w = original_image.shape[:2]
        resized = maskrcnn_resize_u8(original_image, self.min_image_size, self.max_image_size)
        d = self.cfg.SIZE_DIVISIBILITY
        model = self._model(-(-resized.shape[0] // d) * d, -(-resized.shape[1] // d) * d)
        (pred,) = model([resized])  # device front end: the bytes cross PCIe, mean subtraction + padding (to_image_list) run on the GPU
        model.paste_device(h, w, [(w, h)])
        model.sync()
        n = len(pred)
        masks = model.fetch("det.masks", 1)[0, :n]
        out = pred.resize((w, h))
        out.add_field("mask", masks[:, None])
        return out

    def select_top_predictions(self, predictions):
        scores = predictions.get_field("scores")
        keep = np.nonzero(scores > self.confidence_threshold)[0]
        predictions = predictions[keep]
        order = np.argsort(-predictions.get_field("scores"), kind="stable")
        return predictions[order]

    def run_on_opencv_image(self, image):
        predictions = self.select_top_predictions(self.compute_prediction(image))
        return self.overlay(image.copy(), predictions)

    @staticmethod
    def compute_colors_for_labels(labels):
        palette = np.array([2 ** 25 - 1, 2 ** 15 - 1, 2 ** 21 - 1], np.int64)
        return ((np.asarray(labels, np.int64)[:, None] * palette) % 255).astype(np.uint8)

    def overlay(self, image, predictions):
        colors = self.compute_colors_for_labels(predictions.get_field("labels")) if len(predictions) else []
        masks = predictions.get_field("mask") if predictions.has_field("mask") else None
        H, W = image.shape[:2]
        for k in range(len(predictions)):
            c = colors[k].astype(np.float32)
            if masks is not None:
                m = masks[k, 0].astype(bool)
                image[m] = (0.5 * image[m] + 0.5 * c).astype(np.uint8)
            x1, y1, x2, y2 = (int(v) for v in predictions.bbox[k])
            x1, x2 = max(0, min(W - 1, x1)), max(0, min(W - 1, x2))
            y1, y2 = max(0, min(H - 1, y1)), max(0, min(H - 1, y2))
            image[y1:y2 + 1, [x1, x2]] = colors[k]
            image[[y1, y2], x1:x2 + 1] = colors[k]
        return image

    def close(self):
        for m in self._models.values():
            m.close()
        self._models = collections.OrderedDict()


def inference(predictor, images, image_ids=None):
    """engine/inference.py-shaped evaluation loop: images (HxWx3 uint8 BGR) -> COCO-format result list
    (bbox + segm) ready for json.dump (README.md:344-347)."""
    from .coco import maskrcnn_results
    results = []
    for i, im in enumerate(images):
        p = predictor.compute_prediction(im)
        iid = image_ids[i] if image_ids is not None else i
        results += maskrcnn_results(iid, p.bbox, p.get_field("scores"), p.get_field("labels"), p.get_field("mask")[:, 0])
    return results
