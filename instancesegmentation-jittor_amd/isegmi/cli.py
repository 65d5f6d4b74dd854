"""Command-line front ends shaped like the reference's (README.md:243-249 and README.md:344-347):

  python -m isegmi.cli eval --trained_model=weights.npz --score_threshold=0.15 --top_k=15 --image=in.png[:out.png]
  python -m isegmi.cli eval --trained_model=weights.npz --images=in_dir:out_dir [--output_coco_json=dets.json]
  python -m isegmi.cli test_net --config-file cfg.yaml [--images dir] [--output results.json]

`--trained_model` / `MODEL.WEIGHT` take the .npz written by tools/import_pth.py; the literal value `random` uses the
seeded synthetic weights (there is no network to fetch the reference's .pth files).  Images are read with PIL.
Both commands run the batched device pipeline (isegmi.predictor.inference / isegmi.yolact.evaluate): uint8 upload, forward, masks at
the original image size, RLE and record packing on the GPU.  Launched as N processes (RANK / LOCAL_RANK / WORLD_SIZE in the environment,
e.g. by `python -m torch.distributed.run --nproc-per-node N -m isegmi.cli ...`) the batches go round-robin to the ranks and the
records are all-gathered with RCCL; rank 0 writes the COCO json.  Nothing here imports torch.
"""
import argparse
import json
import os
import sys

import numpy as np


def _load_image_bgr(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()


def _save_image_bgr(path, img):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(path)


def _weights(spec, kind, depth=50, ycfg=None):
    from .weights import maskrcnn_state_dict, yolact_state_dict
    if spec in ("", "random", None):
        if kind == "yolact" and ycfg is not None:
            return yolact_state_dict(1234, ycfg.depth, ycfg.num_priors, ycfg.dcn_layers, ycfg.dcn_interval, ycfg.use_maskiou, ycfg.backbone)
        return yolact_state_dict(1234, depth) if kind == "yolact" else maskrcnn_state_dict(1234, depth)
    if not spec.endswith(".npz"):
        raise SystemExit("%s: convert upstream .pth/.pkl with tools/import_pth.py first" % spec)
    return dict(np.load(spec))


def _overlay(img, masks, boxes, classes):
    from .predictor import COCODemo
    colors = COCODemo.compute_colors_for_labels(np.asarray(classes) + 1)
    out = img.copy()
    for k in range(len(classes)):
        m = masks[k].astype(bool)
        out[m] = (0.5 * out[m] + 0.5 * colors[k].astype(np.float32)).astype(np.uint8)
        x1, y1, x2, y2 = (int(v) for v in boxes[k])
        x2, y2 = min(x2, out.shape[1] - 1), min(y2, out.shape[0] - 1)
        out[y1:y2 + 1, [x1, x2]] = colors[k]
        out[[y1, y2], x1:x2 + 1] = colors[k]
    return out


def _rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def _image_sizes(files):
    from PIL import Image
    sizes = []
    for f in files:
        with Image.open(f) as im:  # header only
            sizes.append((im.size[1], im.size[0]))
    return sizes


def cmd_eval(a):
    """Yolact eval.py: evalimage / evalimages (+ Detections.dump when --output_coco_json is given)."""
    from .coco import dump, rle_decode
    from .yolact import Yolact, YolactConfig, evaluate
    # upstream eval.py --config: yolact_resnet50_config (default here), yolact_base_config (R101), yolact_im700_config, and the
    # YOLACT++ pair yolact_plus_resnet50_config / yolact_plus_base_config (DCNv2 backbones, 9 anchors, mask re-scoring)
    cfg = {"yolact_resnet50_config": YolactConfig(), "yolact_base_config": YolactConfig.base(), "yolact_im700_config": YolactConfig.im700(),
           "yolact_plus_resnet50_config": YolactConfig.plus_resnet50(), "yolact_plus_base_config": YolactConfig.plus_base(),
           "yolact_darknet53_config": YolactConfig.darknet53()}[a.config]
    rank, world, local = _rank_world()
    jobs = []
    if a.image:
        src, _, dst = a.image.partition(":")
        jobs.append((src, dst or None))
    if a.images:
        src, _, dst = a.images.partition(":")
        os.makedirs(dst, exist_ok=True)
        for f in sorted(os.listdir(src)):
            jobs.append((os.path.join(src, f), os.path.join(dst, os.path.splitext(f)[0] + ".png")))
    files = [j[0] for j in jobs]
    sizes = _image_sizes(files)
    net = Yolact(_weights(a.trained_model, "yolact", cfg.depth, cfg), cfg, max_batch=a.batch_size, device=local)
    results = evaluate(net, lambda i: _load_image_bgr(files[i]), batch_size=a.batch_size, score_threshold=a.score_threshold, top_k=a.top_k,
                       rank=rank, world=world, sizes=sizes)
    net.close()
    if rank == 0:
        by_img = {}
        for r in results:
            by_img.setdefault(r["image_id"], []).append(r)
        for i, (src, dst) in enumerate(jobs):
            dets = by_img.get(i, [])
            print("%s: %d detections" % (src, len(dets)))
            if dst and dets:  # overlay from the records themselves: RLE -> mask, xywh -> box
                from .coco import COCO_CATEGORY_IDS
                masks = np.stack([rle_decode(d["segmentation"]) for d in dets])
                boxes = [[d["bbox"][0], d["bbox"][1], d["bbox"][0] + d["bbox"][2], d["bbox"][1] + d["bbox"][3]] for d in dets]
                classes = [COCO_CATEGORY_IDS.index(d["category_id"]) for d in dets]
                _save_image_bgr(dst, _overlay(_load_image_bgr(src), masks, boxes, classes))
            elif dst:
                _save_image_bgr(dst, _load_image_bgr(src))
        if a.output_coco_json:
            dump(results, a.output_coco_json)
    return results


def cmd_test_net(a):
    """tools/test_net.py: build the model from the yaml, run inference() over the images, write COCO-format json."""
    from .config import cfg, to_maskrcnn_config
    from .predictor import COCODemo, inference
    c = cfg.clone()
    if a.config_file:
        c.merge_from_file(a.config_file)
    c.merge_from_list(a.opts)
    mc = to_maskrcnn_config(c)
    rank, world, local = _rank_world()
    files = sorted(os.path.join(a.images, f) for f in os.listdir(a.images)) if a.images else []
    demo = COCODemo(mc, min_image_size=mc.MIN_SIZE_TEST, confidence_threshold=0.0, state_dict=_weights(c.MODEL.WEIGHT, "maskrcnn", mc.depth),
                    max_image_size=mc.MAX_SIZE_TEST, device=local, max_batch=a.batch_size)
    stats = {}
    results = inference(demo, lambda i: _load_image_bgr(files[i]), batch_size=a.batch_size, group=a.group, rank=rank, world=world,
                        sizes=_image_sizes(files), stats=stats) if files else []
    demo.close()
    if rank == 0:
        with open(a.output, "w") as f:
            json.dump(results, f)
        print("wrote %d results for %d images to %s (%d steps of %d images on %d rank(s))" % (
            len(results), len(files), a.output, stats.get("steps", 0), a.batch_size, world))
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(prog="isegmi.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    e = sub.add_parser("eval", help="Yolact eval.py-style image evaluation")
    e.add_argument("--trained_model", default="random")
    e.add_argument("--config", default="yolact_resnet50_config", choices=["yolact_resnet50_config", "yolact_base_config", "yolact_im700_config", "yolact_plus_resnet50_config", "yolact_plus_base_config", "yolact_darknet53_config"])
    e.add_argument("--score_threshold", type=float, default=0.0)
    e.add_argument("--top_k", type=int, default=5)
    e.add_argument("--image", default=None, help="in.png[:out.png]")
    e.add_argument("--images", default=None, help="in_dir:out_dir")
    e.add_argument("--output_coco_json", default=None)
    e.add_argument("--batch_size", type=int, default=8, help="images per step and rank")
    t = sub.add_parser("test_net", help="detectron tools/test_net.py-style evaluation -> COCO json")
    t.add_argument("--config-file", dest="config_file", default="")
    t.add_argument("--images", default=None, help="directory of images (the COCO dataset catalog is out of scope)")
    t.add_argument("--output", default="results.json")
    t.add_argument("--batch_size", type=int, default=2, help="images per step and rank (TEST.IMS_PER_BATCH / ranks)")
    t.add_argument("--group", default="canvas", choices=["canvas", "aspect"],
                   help="canvas: batch only images with the same padded canvas (every result equals the single-image result); aspect: upstream's ASPECT_RATIO_GROUPING")
    t.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    a = ap.parse_args(argv)
    return cmd_eval(a) if a.cmd == "eval" else cmd_test_net(a)


if __name__ == "__main__":
    main()
