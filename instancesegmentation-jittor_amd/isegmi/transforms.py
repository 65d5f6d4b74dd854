"""Real-image front end: the host-side transforms either side of the hot path.

Mask R-CNN: COCODemo.build_transform = Resize(min 800, max 1333, PIL bilinear) -> BGR 0..255 -> subtract
PIXEL_MEAN (M1; README.md:320-331 passes an HxWx3 uint8 BGR array).  Yolact: FastBaseTransform resizes to
550x550 (bilinear, align_corners=False, no aspect keep), normalises with BGR mean/std and flips to RGB (Y1).
"""
import numpy as np


def get_size(w, h, min_size=800, max_size=1333):
    """maskrcnn-benchmark transforms.Resize.get_size -> (oh, ow)."""
    size = min_size
    mn, mx = float(min(w, h)), float(max(w, h))
    if mx / mn * size > max_size:
        size = int(round(max_size * mn / mx))
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        ow = size
        oh = int(size * h / w)
    else:
        oh = size
        ow = int(size * w / h)
    return oh, ow


def maskrcnn_resize_u8(image_bgr_u8, min_size=800, max_size=1333):
    """HxWx3 uint8 BGR -> resized uint8 BGR (PIL bilinear, as torchvision F.resize on a PIL image; PIL rounds back to bytes)."""
    from PIL import Image
    h, w = image_bgr_u8.shape[:2]
    oh, ow = get_size(w, h, min_size, max_size)
    rgb = Image.fromarray(np.ascontiguousarray(image_bgr_u8[:, :, ::-1]))
    rgb = rgb.resize((ow, oh), Image.BILINEAR)
    return np.ascontiguousarray(np.asarray(rgb, np.uint8)[:, :, ::-1])


def maskrcnn_resize(image_bgr_u8, min_size=800, max_size=1333):
    """The same as float32 BGR 0..255 (the input of prepare_images on the host path)."""
    return maskrcnn_resize_u8(image_bgr_u8, min_size, max_size).astype(np.float32)


def bilinear_resize(img, oh, ow):
    """F.interpolate(mode='bilinear', align_corners=False) on an HxWxC float array (numpy, host side)."""
    img = np.asarray(img, np.float32)
    h, w = img.shape[:2]

    def coef(o, i):
        src = np.maximum((np.arange(o, dtype=np.float32) + 0.5) * np.float32(i / o) - 0.5, 0)
        i0 = np.minimum(src.astype(np.int64), i - 1)
        i1 = np.minimum(i0 + 1, i - 1)
        l1 = (src - i0).astype(np.float32)
        return i0, i1, 1 - l1, l1
    y0, y1, ly0, ly1 = coef(oh, h)
    x0, x1, lx0, lx1 = coef(ow, w)
    top = img[y0][:, x0] * lx0[None, :, None] + img[y0][:, x1] * lx1[None, :, None]
    bot = img[y1][:, x0] * lx0[None, :, None] + img[y1][:, x1] * lx1[None, :, None]
    return top * ly0[:, None, None] + bot * ly1[:, None, None]


def yolact_transform(image_bgr_u8, size=550, darknet=False):
    """FastBaseTransform: resize to size x size, then the backbone's transform (ResNet: mean / std; darknet=True: x / 255), RGB."""
    from .yolact import darknet_base_transform, fast_base_transform
    x = bilinear_resize(np.asarray(image_bgr_u8, np.float32), size, size)[None]
    return darknet_base_transform(x) if darknet else fast_base_transform(x)
