"""Mask R-CNN end to end: HIP engine vs the numpy/C oracle model, same seeded weights and images.
Backbone/FPN features, proposals, detections (boxes, scores, labels), 28x28 masks and pasted masks bit-exact."""
import numpy as np
import pytest

from oracle.maskrcnn_ref import MaskRCNNRef

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    from isegmi.weights import maskrcnn_state_dict
    return maskrcnn_state_dict(1234)


def test_anchor_generators_agree():
    from isegmi.maskrcnn import generate_anchors, grid_anchors
    from oracle.maskrcnn_ref import cell_anchors, grid_anchors as ref_grid
    for stride, size in zip((4, 8, 16, 32, 64), (32, 64, 128, 256, 512)):
        a = generate_anchors(stride, size, (0.5, 1.0, 2.0)); b = cell_anchors(stride, size)
        assert np.array_equal(a, b)
        assert np.array_equal(grid_anchors(5, 7, stride, a), ref_grid(5, 7, stride, b))
    assert np.array_equal(generate_anchors(4, 32, (0.5, 1.0, 2.0)), np.array([[-22, -10, 25, 13], [-14, -14, 17, 17], [-10, -22, 13, 25]], np.float32))


def test_maskrcnn_small_batch2_bit_exact(ffi, sd):
    from isegmi.maskrcnn import MaskRCNN, prepare_images
    rng = np.random.default_rng(20261003)
    imgs = [rng.uniform(0, 255, (250, 340, 3)).astype(np.float32), rng.uniform(0, 255, (256, 300, 3)).astype(np.float32)]
    x, hw = prepare_images(imgs)
    assert x.shape == (2, 256, 352, 3)
    model = MaskRCNN(sd, x.shape[1], x.shape[2], max_batch=2)
    out = model(x, hw)
    ref = MaskRCNNRef(sd)
    rd = ref.forward(x, hw)
    for name in ("P2", "P3", "P4", "P5", "P6"):
        assert np.array_equal(model.fetch(name, 2), ref.feats[name]), name
    pc = model.fetch("proposal_count", 2); pr = model.fetch("proposals", 2); ps = model.fetch("proposal_scores", 2)
    total = 0
    for n in range(2):
        r = rd[n]
        assert pc[n] == len(r["proposals"])
        assert np.array_equal(ps[n, : pc[n]], r["proposal_scores"]) and np.array_equal(pr[n, : pc[n]], r["proposals"])
        bl = out[n]
        assert len(bl) == len(r["score"])
        assert np.array_equal(bl.get_field("labels"), r["label"].astype(np.int64))
        assert np.array_equal(bl.get_field("scores"), r["score"]) and np.array_equal(bl.bbox, r["box"])
        assert np.array_equal(bl.get_field("mask")[:, 0], r["mask28"])
        total += len(bl)
    assert total > 20
    # paste at the network input size and at a resized "original" size
    oh, ow = 256, 352
    model.paste_device(oh, ow); model.sync()
    masks = model.fetch("det.masks", 2)
    for n in range(2):
        rm, _ = MaskRCNNRef.paste(rd[n], oh, ow)
        assert np.array_equal(masks[n, : len(rm)], rm)
    orig = np.array([[680, 500], [600, 512]])  # (w, h) originals
    model.paste_device(520, 700, orig); model.sync()
    masks = model.fetch("det.masks", 2); rb = model.fetch("det.box_resized", 2)
    for n in range(2):
        ratio = (np.float32(orig[n, 0] / hw[n, 1]), np.float32(orig[n, 1] / hw[n, 0]))
        rm, rbox = MaskRCNNRef.paste(rd[n], 520, 700, ratio)
        assert np.array_equal(rb[n, : len(rm)], rbox) and np.array_equal(masks[n, : len(rm)], rm)
    model.close()
