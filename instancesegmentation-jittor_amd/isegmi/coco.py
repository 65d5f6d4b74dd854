"""COCO-format results: bbox xywh, category-id remap and RLE mask encoding.

The reference reaches this through `python tools/test_net.py` -> inference() -> COCO json (README.md:344-347)
and Yolact eval.py's Detections.add_bbox/add_mask/dump (README.md:243-249); annotation layout README.md:55-66.
pycocotools is not in the image, so the run-length encoder and its LEB128-style string compression
(pycocotools maskApi.c rleEncode / rleToString / rleFrString) are restated here in numpy/Python.
"""
import json

import numpy as np

# contiguous label 1..80 -> COCO category id (the 2017 "thing" classes)
COCO_CATEGORY_IDS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 27, 28, 31, 32, 33, 34,
                     35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65,
                     67, 70, 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 84, 85, 86, 87, 88, 89, 90]
COCO_CLASSES = ("person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant",
                "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra", "giraffe",
                "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite", "baseball bat",
                "baseball glove", "skateboard", "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl",
                "banana", "apple", "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
                "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard", "cell phone", "microwave",
                "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors", "teddy bear", "hair drier", "toothbrush")


def rle_counts(mask):
    """Column-major run lengths of a binary HxW mask, starting with the count of zeros (rleEncode)."""
    m = np.asarray(mask).astype(bool)
    flat = m.ravel(order="F")
    if flat.size == 0:
        return [0]
    change = np.nonzero(flat[1:] != flat[:-1])[0] + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(bounds).tolist()
    if flat[0]:
        counts = [0] + counts
    return counts


def rle_to_string(counts):
    """pycocotools rleToString: delta vs the count two runs back, 5 data bits + continuation bit per char."""
    out = []
    for i, x in enumerate(counts):
        x = int(x)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            c = x & 0x1F
            x >>= 5
            more = (x != -1) if (c & 0x10) else (x != 0)
            if more:
                c |= 0x20
            out.append(chr(c + 48))
    return "".join(out)


def rle_from_string(s):
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_encode(mask):
    h, w = np.asarray(mask).shape
    return {"size": [int(h), int(w)], "counts": rle_to_string(rle_counts(mask))}


def rle_decode(rle):
    h, w = rle["size"]
    counts = rle_from_string(rle["counts"]) if isinstance(rle["counts"], str) else list(rle["counts"])
    flat = np.zeros(h * w, np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape((h, w), order="F")


def maskrcnn_results(image_id, boxes_xyxy, scores, labels, masks=None):
    """maskrcnn-benchmark prepare_for_coco_detection/segmentation: bbox xywh with the legacy +1 widths,
    category_id via the contiguous->json id map, segmentation = RLE of the pasted HxW mask."""
    out = []
    b = np.asarray(boxes_xyxy, np.float32).reshape(-1, 4)
    for k in range(b.shape[0]):
        x1, y1, x2, y2 = (float(v) for v in b[k])
        d = {"image_id": int(image_id), "category_id": COCO_CATEGORY_IDS[int(labels[k]) - 1],
             "bbox": [x1, y1, x2 - x1 + 1.0, y2 - y1 + 1.0], "score": float(scores[k])}
        if masks is not None:
            d["segmentation"] = rle_encode(masks[k])
        out.append(d)
    return out


def yolact_results(image_id, classes, scores, boxes_xyxy_int, masks=None, mask_scores=None):
    """Yolact eval.py Detections.add_bbox/add_mask: bbox [x1,y1,w,h] rounded to 0.1, class 0..79 -> COCO id.
    mask_scores (YOLACT++ re-scoring): upstream writes the mask entry with the re-scored value; kept here as "mask_score"."""
    out = []
    b = np.asarray(boxes_xyxy_int).reshape(-1, 4)
    for k in range(b.shape[0]):
        x1, y1, x2, y2 = (float(v) for v in b[k])
        bbox = [round(v * 10) / 10 for v in (x1, y1, x2 - x1, y2 - y1)]
        d = {"image_id": int(image_id), "category_id": COCO_CATEGORY_IDS[int(classes[k])], "bbox": bbox, "score": float(scores[k])}
        if masks is not None:
            d["segmentation"] = rle_encode(masks[k])
        if mask_scores is not None:
            d["mask_score"] = float(mask_scores[k])
        out.append(d)
    return out


def results_from_records(rec, image_ids, image_hw, kind, K=100, score_threshold=None, top_k=None):
    """One rank's unpacked record block (isegmi.dist.unpack_coco_records: the device-side output of a step) -> COCO result dicts, the
    same records maskrcnn_results / yolact_results build from host arrays: bbox conversion and category map here, `segmentation.counts`
    straight from the device-made strings.  image_ids / image_hw: per image slot of the batch (None id = an empty padding slot).
    score_threshold / top_k (Yolact eval.py --score_threshold / --top_k): keep score > threshold, then the first top_k (score order)."""
    out = []
    count, box, score, label, so, chars = rec["count"], rec["box"], rec["score"], rec["label"], rec["str_off"], rec["chars"]
    ms = rec.get("mscore")
    for n, iid in enumerate(image_ids):
        c = int(count[n])
        if iid is None or c == 0:
            continue
        h, w = int(image_hw[n][0]), int(image_hw[n][1])
        b = np.asarray(box[n, :c], np.float64)
        if kind == 2:   # prepare_for_coco_detection: xyxy -> xywh with the legacy +1
            bb = np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0] + 1.0, b[:, 3] - b[:, 1] + 1.0], 1).tolist()
            cats = [COCO_CATEGORY_IDS[int(l) - 1] for l in label[n, :c]]
        else:           # Detections.add_bbox: [x1, y1, w, h] rounded to 0.1
            bb = (np.round(np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1) * 10) / 10).tolist()
            cats = [COCO_CATEGORY_IDS[int(l)] for l in label[n, :c]]
        sc = np.asarray(score[n, :c], np.float64).tolist()
        offs = so[n * K:n * K + c + 1].tolist()
        keep = range(c)
        if score_threshold is not None:
            keep = [k for k in keep if score[n, k] > np.float32(score_threshold)]
        if top_k is not None:
            keep = list(keep)[:top_k]
        for k in keep:
            d = {"image_id": int(iid), "category_id": cats[k], "bbox": bb[k], "score": sc[k],
                 "segmentation": {"size": [h, w], "counts": chars[offs[k]:offs[k + 1]].decode("ascii")}}
            if ms is not None:
                d["mask_score"] = float(ms[n, k])
            out.append(d)
    return out


def dump(results, path):
    with open(path, "w") as f:
        json.dump(results, f)
