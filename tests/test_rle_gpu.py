"""Device-side COCO run-length encoding (isegmi_op_rle_encode) against the CPU oracle's restatement of pycocotools rleEncode / rleToString:
run counts and strings bit-identical, for ragged shapes (h, w not multiples of 64 / 256), empty / full / noisy masks, invalid slots and
per-image windows inside a common plane."""
import numpy as np
import pytest

from oracle import ora

pytestmark = pytest.mark.gpu


def _blobs(rng, h, w, n):
    yy, xx = np.mgrid[0:h, 0:w]
    m = np.zeros((h, w), bool)
    for _ in range(n):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        ry, rx = rng.uniform(2, h / 2 + 2), rng.uniform(2, w / 2 + 2)
        m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0
    return m


def _check(ffi, masks, count=None, image_hw=None, **kw):
    N, K, h, w = masks.shape
    ro, cn, so, ch, st = ffi.rle_encode(masks, count, image_hw, **kw)
    assert st[2] == 0, "overflow flagged"
    total_runs = total_chars = 0
    for n in range(N):
        hi, wi = (h, w) if image_hw is None else (int(image_hw[n][0]), int(image_hw[n][1]))
        for k in range(K):
            m = n * K + k
            valid = count is None or k < int(count[n])
            if not valid:
                assert ro[m + 1] == ro[m] and so[m + 1] == so[m], (n, k)
                continue
            ref = ora.rle_encode(masks[n, k, :hi, :wi])
            got = cn[ro[m]:ro[m + 1]]
            assert np.array_equal(got, ref), (n, k, got[:8], ref[:8])
            s = ch[so[m]:so[m + 1]].decode("ascii")
            assert s == ora.rle_to_string(ref), (n, k)
            total_runs += len(ref); total_chars += len(s)
    assert st[0] == total_runs == ro[-1] and st[1] == total_chars == so[-1]


@pytest.mark.parametrize("h,w", [(1, 1), (3, 5), (64, 64), (65, 63), (138, 138), (200, 257), (550, 550), (480, 640)])
def test_rle_matches_oracle_shapes(ffi, h, w):
    rng = np.random.default_rng(h * 1000 + w)
    K = 7
    masks = np.zeros((2, K, h, w), np.uint8)
    masks[0, 1] = 1                                              # full: counts [0, h*w]
    masks[0, 2] = rng.uniform(0, 1, (h, w)) < 0.5                # noise: a run start at almost every pixel
    masks[0, 3] = _blobs(rng, h, w, 3)
    masks[0, 4, :, 0] = 1                                        # first column set: leading 0 count
    masks[0, 5, -1, -1] = 1                                      # only the very last pixel
    masks[0, 6] = (np.add.outer(np.arange(h), np.arange(w)) % 2)  # checkerboard: h*w runs
    for k in range(K):
        masks[1, k] = _blobs(rng, h, w, 1 + k) * (rng.uniform(0, 1, (h, w)) < 0.97)
    _check(ffi, masks)


def test_rle_invalid_slots_and_windows(ffi):
    rng = np.random.default_rng(5)
    N, K, h, w = 3, 6, 130, 300
    masks = (rng.uniform(0, 1, (N, K, h, w)) < 0.3).astype(np.uint8)
    masks[:, :, 40:90, 100:200] = 1
    count = np.array([0, 6, 3], np.int32)            # image 0 has no detections at all (its slots precede every run)
    _check(ffi, masks, count)
    hw = np.array([[130, 300], [64, 299], [129, 1]], np.int32)  # per-image windows: only the top-left (h_n, w_n) of each plane is encoded
    _check(ffi, masks, count, hw)
    _check(ffi, masks, np.array([2, 0, 0], np.int32), hw)       # trailing images without detections
    _check(ffi, masks, np.array([0, 0, 0], np.int32))           # nothing at all


def test_rle_overflow_is_flagged(ffi):
    masks = (np.add.outer(np.arange(64), np.arange(64)) % 2).astype(np.uint8)[None, None]
    nruns = len(ora.rle_encode(masks[0, 0]))                                         # 4033: a column's last pixel equals the next column's first
    ro, cn, so, ch, st = ffi.rle_encode(masks, cap_runs=1024, cap_chars=4096)
    assert st[2] & 1 and st[0] == nruns > 1024
    ro, cn, so, ch, st = ffi.rle_encode(masks, cap_runs=8192, cap_chars=100)        # runs fit, characters do not
    assert st[2] == 2 and st[0] == nruns


def test_rle_mask_rcnn_sized_planes(ffi):
    """Mask R-CNN sized planes (800 x 1333) with box-shaped content; deterministic output bytes."""
    rng = np.random.default_rng(9)
    masks = np.zeros((1, 5, 800, 1333), np.uint8)
    for k in range(5):
        y0, x0 = int(rng.integers(0, 600)), int(rng.integers(0, 1000))
        masks[0, k, y0:y0 + 200, x0:x0 + 333] = _blobs(rng, 200, 333, 4)
    a = ffi.rle_encode(masks, np.array([5], np.int32))
    b = ffi.rle_encode(masks, np.array([5], np.int32))
    assert all(np.array_equal(np.asarray(x), np.asarray(y)) if not isinstance(x, bytes) else x == y for x, y in zip(a, b))
    _check(ffi, masks, np.array([5], np.int32))


def test_rle_reads_only_the_windows(ffi):
    """with per-slot windows (what the paste / mask assembly kernels leave next to the planes) nothing outside a window is read: the planes
    carry GARBAGE there and the result still equals the encoding of the clean masks -- windows touching every border, single pixels, a
    window ending exactly on a 64-row chunk boundary, an empty window"""
    rng = np.random.default_rng(21)
    N, K, h, w = 2, 9, 200, 150
    clean = np.zeros((N, K, h, w), np.uint8)
    wins = np.zeros((N, K, 4), np.int32)
    boxes = [(10, 20, 90, 130), (0, 0, 150, 200), (0, 60, 40, 200), (100, 0, 150, 64), (30, 64, 31, 128), (149, 199, 150, 200), (0, 0, 1, 1), (5, 5, 5, 5), (20, 100, 60, 192)]
    for n in range(N):
        for k, (x0, y0, x1, y1) in enumerate(boxes):
            wins[n, k] = (x0, y0, x1, y1)
            if x1 > x0 and y1 > y0:
                clean[n, k, y0:y1, x0:x1] = (rng.uniform(0, 1, (y1 - y0, x1 - x0)) < (0.5 if (n + k) % 2 else 0.95)).astype(np.uint8)
    clean[0, 1] = 1                                   # full plane: window = the image, every column ends set
    clean[1, 2, 60:200, 0:40] = 1                     # solid block touching the left and bottom borders
    dirty = clean.copy()
    for n in range(N):
        for k, (x0, y0, x1, y1) in enumerate(boxes):
            g = rng.integers(1, 255, (h, w)).astype(np.uint8)
            g[y0:y1, x0:x1] = clean[n, k, y0:y1, x0:x1]
            dirty[n, k] = g
    count = np.array([9, 9], np.int32)
    ro, cn, so, ch, st = ffi.rle_encode(dirty, count, None, windows=wins)
    assert st[2] == 0
    for n in range(N):
        for k in range(K):
            m = n * K + k
            ref = ora.rle_encode(clean[n, k])
            assert np.array_equal(cn[ro[m]:ro[m + 1]], ref), (n, k, cn[ro[m]:ro[m + 1]][:6], ref[:6])
            assert ch[so[m]:so[m + 1]].decode("ascii") == ora.rle_to_string(ref)
    # windows + per-image sizes smaller than the plane (the window is clamped to the image)
    hw = np.array([[180, 120], [200, 150]], np.int32)
    ro, cn, so, ch, st = ffi.rle_encode(dirty, count, hw, windows=wins)
    for n in range(N):
        for k in range(K):
            m = n * K + k
            ref = ora.rle_encode(clean[n, k, :hw[n, 0], :hw[n, 1]])
            assert np.array_equal(cn[ro[m]:ro[m + 1]], ref), (n, k)
