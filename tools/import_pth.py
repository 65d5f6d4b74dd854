"""Pretrained-weight importer: upstream .pth/.pkl state dict -> .npz usable by
isegmi.yolact.Yolact / isegmi.maskrcnn.MaskRCNN (which already consume upstream key names).

usage: python tools/import_pth.py weights/yolact_resnet50_54_800000.pth out.npz
torch is used HERE only to unpickle (a tool, not the product); `num_batches_tracked` and optimizer state are dropped,
a leading 'module.' (DataParallel) and a top-level 'model' key (maskrcnn-benchmark checkpoints) are stripped.
"""
import sys

import numpy as np


def convert(sd):
    if "model" in sd and isinstance(sd["model"], dict):
        sd = sd["model"]
    out = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked") or not hasattr(v, "shape"):
            continue
        k = k[7:] if k.startswith("module.") else k
        out[k] = np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, np.float32)
    return out


def main():
    import torch
    sd = torch.load(sys.argv[1], map_location="cpu")
    out = convert(sd)
    np.savez(sys.argv[2], **out)
    print("wrote %d tensors, %.1f M parameters" % (len(out), sum(v.size for v in out.values()) / 1e6))


if __name__ == "__main__":
    main()
