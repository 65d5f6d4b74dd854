"""COCODemo -- the predictor object of README.md:288-335, backed by libisegmi.

    coco_demo = COCODemo(cfg, min_image_size=800, confidence_threshold=0.5, state_dict=sd)
    predictions = coco_demo.run_on_opencv_image(image)     # image: HxWx3 uint8 BGR -> annotated HxWx3 uint8

`cfg` is a MaskRCNNConfig (the handful of inference constants the path needs, keyed like the yaml).
compute_prediction / select_top_predictions follow the upstream demo ([UPSTREAM-RECALL]);
overlay drawing is numpy + PIL (cv2 is not in the image): mask tint, box outline and the "class score" label upstream's
overlay_class_names writes at the box's top-left corner (PIL's built-in font instead of cv2.putText's Hershey face).
"""
import numpy as np

from .coco import COCO_CLASSES
from .maskrcnn import BoxList, MaskRCNN, MaskRCNNConfig
from .transforms import maskrcnn_resize_u8


class COCODemo:
    CATEGORIES = ("__background",) + COCO_CLASSES

    def __init__(self, cfg=None, min_image_size=800, confidence_threshold=0.5, state_dict=None, max_image_size=1333, device=0, max_batch=2,
                 fp16=False):
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
        self.state_dict, self.device, self.max_batch, self.fp16 = state_dict, device, int(max_batch), bool(fp16)
        self._model = None

    def engine(self, max_batch=None):
        """THE engine of this predictor: weights packed once, every buffer sized once for the largest canvas the resize rule can produce
        (long side max_image_size, either orientation, rounded up to SIZE_DIVISIBILITY).  Each batch then runs on its own padded canvas,
        exactly as upstream's to_image_list pads it (results depend on the canvas: the RPN sees the padding)."""
        want = int(max_batch or self.max_batch)
        if self._model is not None and self._model.max_batch < want:
            self._model.close()
            self._model = None
        if self._model is None:
            d = self.cfg.SIZE_DIVISIBILITY
            m = -(-max(self.max_image_size, self.min_image_size) // d) * d
            self._model = MaskRCNN(self.state_dict, m, m, cfg=self.cfg, max_batch=want, device=self.device, fp16=self.fp16)
            self.memory = self._model.reserve()  # (weight bytes, activation / workspace bytes): the predictor's peak device memory
        return self._model

    def compute_prediction(self, original_image):
        """-> BoxList in ORIGINAL image coordinates with scores, labels and mask [n,1,H,W] uint8 (Masker output)."""
        h, w = original_image.shape[:2]
        resized = maskrcnn_resize_u8(original_image, self.min_image_size, self.max_image_size)
        model = self.engine()
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
        return self.overlay_class_names(image, predictions)

    def overlay_class_names(self, image, predictions):
        """demo/predictor.py overlay_class_names: "<class name>: <score>" in white at the top-left corner of every box (README.md:331-334:
        the image run_on_opencv_image returns carries boxes, masks AND labels)."""
        if len(predictions) == 0 or not predictions.has_field("scores"):
            return image
        from PIL import Image, ImageDraw
        pil = Image.fromarray(np.ascontiguousarray(image[:, :, ::-1]))   # BGR -> RGB for PIL
        draw = ImageDraw.Draw(pil)
        scores, labels = predictions.get_field("scores").tolist(), predictions.get_field("labels").tolist()
        for box, score, label in zip(predictions.bbox, scores, labels):
            x, y = int(box[0]), int(box[1])
            draw.text((max(0, x), max(0, y)), "%s: %.2f" % (self.CATEGORIES[int(label)], score), fill=(255, 255, 255))
        image[...] = np.asarray(pil, np.uint8)[:, :, ::-1]
        return image

    def close(self):
        if self._model is not None:
            self._model.close()
            self._model = None


def inference(predictor, images, image_ids=None, batch_size=None, group="canvas", rank=0, world=1, sizes=None, stats=None, force_gather=False,
              workers=4):
    """engine/inference.py-shaped evaluation (README.md:344-347): images -> COCO-format result list (bbox + segm) ready for json.dump.

    The path the benchmark measures, end to end: batches of `batch_size` resized uint8 images go up through pinned memory (double-buffered),
    mean subtraction / padding, the forward, Masker paste at every image's ORIGINAL size, pycocotools RLE and the packing of one fixed-size
    record block all run on the device; the block comes back asynchronously (one rank) or is all-gathered over RCCL (`world` ranks: the
    batches of the global schedule go round-robin to the ranks) while the next batch computes.  Every rank returns the complete
    list, ordered by image, detections in the engine's output order.

    images: sequence of HxWx3 uint8 BGR arrays, or a callable i -> array together with sizes = [(h, w), ...] (lazy loading).
    group:  "canvas" -- batch only images whose padded network canvas is identical, so every image's result equals its single-image
            result (to_image_list pads a batch to its largest member and the RPN sees the padding); "aspect" -- upstream's
            ASPECT_RATIO_GROUPING (portrait / landscape), results then depend on the batch composition exactly as upstream's do.
    stats:  optional dict, receives steps / images / seconds of the device loop.
    workers: host threads that load + resize (PIL, which releases the GIL) the NEXT batch while the current one is enqueued and the previous
            one's records are unpacked; 0 = inline."""
    import time
    from .coco import results_from_records
    from .pipeline import record_capacity, run_record_loop, schedule_batches
    from .transforms import get_size
    from .maskrcnn import padded_canvas
    from . import _ffi
    get = images if callable(images) else images.__getitem__
    if sizes is None:
        if callable(images):
            raise ValueError("inference(): a callable image source needs sizes=[(h, w), ...]")
        sizes = [im.shape[:2] for im in images]
    n_img = len(sizes)
    ids = list(image_ids) if image_ids is not None else list(range(n_img))
    bs = int(batch_size or predictor.max_batch)
    model = predictor.engine(bs)
    cfg = predictor.cfg
    rs = [get_size(w, h, predictor.min_image_size, predictor.max_image_size) for h, w in sizes]   # resized (h, w) per image
    if group == "canvas":
        keys = [padded_canvas([r], cfg.SIZE_DIVISIBILITY) for r in rs]
    elif group == "aspect":
        keys = [int(h >= w) for h, w in sizes]
    else:
        raise ValueError("group: 'canvas' or 'aspect'")
    batches = schedule_batches(keys, bs)
    K = record_capacity(model)
    pin = [_ffi.PinnedBuffer((bs * model.H * model.W * 3,), np.uint8) for _ in range(2)]
    for slot in (0, 1):
        model._u8_staging(slot, bs * model.H * model.W * 3)   # device staging at its final size: no sync + re-allocation on a larger batch
    per_image = [None] * n_img

    def consume(step, recs):
        for r, rec in enumerate(recs):       # rank r ran batch step * world + r of the global schedule
            j = step * world + r
            if j >= len(batches):
                continue
            b = batches[j]
            slot_ids = [ids[i] for i in b] + [None] * (bs - len(b))
            slot_hw = [sizes[i] for i in b] + [(1, 1)] * (bs - len(b))
            res = results_from_records(rec, slot_ids, slot_hw, 2, K)
            by_id = {}
            for d in res:
                by_id.setdefault(d["image_id"], []).append(d)
            for i in b:
                per_image[i] = by_id.get(ids[i], [])

    def load(i):   # host: decode + PIL resize (outside the hot path)
        return maskrcnn_resize_u8(get(i), predictor.min_image_size, predictor.max_image_size)
    pool = None
    if workers and workers > 0:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=int(workers))
    mine = [j for j in range(rank, len(batches), world)]
    prefetch = {}

    def request(j):
        if j is not None and j < len(batches) and j not in prefetch:
            prefetch[j] = [pool.submit(load, i) for i in batches[j]] if pool else None

    def enqueue(step, slot):
        j = step * world + rank
        if j >= len(batches):
            return False
        b = batches[j]
        request(j + world)                   # the next batch of this rank resizes on the worker threads meanwhile
        futs = prefetch.pop(j, None)         # (a step that is redone after an RLE overflow loads its images again, inline)
        resized = [f.result() for f in futs] if futs else [load(i) for i in b]
        off, hw = 0, []
        for im in resized:                   # into pinned memory, back to back
            pin[slot].array[off:off + im.size] = im.reshape(-1)
            off += im.size
            hw.append(im.shape[:2])
        model.upload_u8_async(pin[slot], hw, slot)
        model.forward_device(len(b), slot)
        oh = [sizes[i] for i in b]
        model.paste_device(max(h for h, _ in oh), max(w for _, w in oh), [(w, h) for h, w in oh])
        model.rle_device(oh)
        return True

    t0 = time.perf_counter()
    nsteps = -(-len(batches) // world)
    try:
        if mine:
            request(mine[0])
        # every way out of the loop -- a bad image, an RLE overflow that does not settle, a ctypes error -- restores the engine (whole mask planes
        # again: `sparse_masks`), frees the pinned buffers, stops the worker threads and closes the gather (run_record_loop's own finally)
        run_record_loop(model, bs, nsteps, enqueue, consume, rank, world, force_gather)
    finally:
        if pool:
            pool.shutdown(wait=True, cancel_futures=True)
        for p in pin:
            p.free()
    if stats is not None:
        stats.update(steps=nsteps, images=n_img, seconds=time.perf_counter() - t0, batches=len(batches), batch_size=bs, world=world)
    return [d for r in per_image if r for d in r]
