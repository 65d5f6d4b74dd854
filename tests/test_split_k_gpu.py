"""`conv_split_k` -- the engines' opt-in latency numerics mode: the backbone's small-M / large-K bottleneck convolutions run on conv tile 15,
four k-ordered partial chains over equal ranges of the K chunks added left to right, so that a 16 x 16 output tile is four short dependent MFMA chains on the
four waves of a SIMD instead of one long one.  It is another valid fp32 evaluation of the same sums and the numerics CONTRACT moves with it: the oracle models
take the same switch (the same layers by the same shape rule, `ora.conv2d(ksplit=4)`) and the engines must equal them bit for bit -- every tensor, index, score,
box and mask -- while the default mode stays what it was.  (Operator level: tests/test_conv_gpu.py::test_conv_split_k_*; the oracle's side by hand:
tests/test_oracle_cpu.py::test_conv_split_k_is_the_stated_sum.)"""
import dataclasses

import numpy as np
import pytest

from oracle.maskrcnn_ref import MaskRCNNRef
from oracle.yolact_ref import YolactRef

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [1, 2])
def test_yolact_split_k_engine_equals_split_k_oracle(ffi, N):
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, YolactConfig, fast_base_transform, postprocess
    sd = yolact_state_dict(1234)
    rng = np.random.default_rng(20261003)
    x = fast_base_transform(rng.uniform(0, 255, (N, 200, 200, 3)).astype(np.float32))
    outs = {}
    for split in (0, 1):
        for groups in ((1, 0) if split else (1,)):   # grouped and per-layer launches: the rule is by layer and shape, never by launch form
            net = Yolact(sd, dataclasses.replace(YolactConfig(), conv_split_k=split), max_batch=N, input_size=200)
            net.set_param("conv_groups", float(groups))
            out = net(x)
            ref = YolactRef(sd, max_size=550, conv_split_k=split)
            refd = ref.forward(x)
            for name in ("C3", "P3", "P5", "proto"):
                got = net.fetch({"C3": "res3.C"}.get(name, name), N)
                assert np.array_equal(got.reshape(ref.feats[name].shape), ref.feats[name]), (split, groups, name)
            for i in range(N):
                d, r = out[i]["detection"], refd[i]
                assert d is not None and len(r["score"]) > 0
                for a, b in (("prior", "prior"), ("class", "cls"), ("score", "score"), ("box", "box"), ("mask", "mask")):
                    assert np.array_equal(d[a], r[b]), (split, groups, a)
            cls, sc, boxes, masks = postprocess(out, 200, 200)
            rc, rs, rb, rm = YolactRef.postprocess(refd[0], 200, 200)
            assert np.array_equal(masks, rm) and np.array_equal(boxes, rb)
            outs[(split, groups)] = net.fetch("P5", N).copy()
            net.close()
    assert not np.array_equal(outs[(0, 1)], outs[(1, 1)])      # the mode changes low bits ...
    assert np.array_equal(outs[(1, 1)], outs[(1, 0)])          # ... and does not depend on the launch form
    assert np.allclose(outs[(0, 1)], outs[(1, 1)], rtol=0, atol=1e-4 * np.abs(outs[(0, 1)]).max())


def test_maskrcnn_split_k_engine_equals_split_k_oracle(ffi):
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig, prepare_images
    from isegmi.weights import maskrcnn_state_dict
    sd = maskrcnn_state_dict(1234)
    rng = np.random.default_rng(20261003)
    x, hw = prepare_images([rng.uniform(0, 255, (250, 340, 3)).astype(np.float32)])
    model = MaskRCNN(sd, x.shape[1], x.shape[2], cfg=dataclasses.replace(MaskRCNNConfig(), CONV_SPLIT_K=1), max_batch=1)
    out = model(x, hw)
    ref = MaskRCNNRef(sd, conv_split_k=1)
    rd = ref.forward(x, hw)
    for name in ("P2", "P5", "P6"):
        assert np.array_equal(model.fetch(name, 1), ref.feats[name]), name
    r, bl = rd[0], out[0]
    pc = model.fetch("proposal_count", 1); pr = model.fetch("proposals", 1)
    assert pc[0] == len(r["proposals"]) and np.array_equal(pr[0, : pc[0]], r["proposals"])
    assert len(bl) == len(r["score"]) > 10
    assert np.array_equal(bl.get_field("labels"), r["label"].astype(np.int64))
    assert np.array_equal(bl.get_field("scores"), r["score"]) and np.array_equal(bl.bbox, r["box"])
    assert np.array_equal(bl.get_field("mask")[:, 0], r["mask28"])
    model.paste_device(x.shape[1], x.shape[2]); model.sync()
    rm, _ = MaskRCNNRef.paste(r, x.shape[1], x.shape[2])
    assert np.array_equal(model.fetch("det.masks", 1)[0, : len(rm)], rm)
    p5_split = model.fetch("P5", 1).copy()
    model.close()
    base = MaskRCNN(sd, x.shape[1], x.shape[2], max_batch=1)
    base(x, hw)
    p5 = base.fetch("P5", 1)
    base.close()
    assert not np.array_equal(p5, p5_split) and np.allclose(p5, p5_split, rtol=0, atol=1e-4 * np.abs(p5).max())


def test_split_k_is_off_by_default_and_fp16_ignores_it(ffi):
    """the default engines never take tile 15 (bs = 8 / bs = 2 numerics and every other test are untouched); an fp16 engine has no split form"""
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, YolactConfig, fast_base_transform
    sd = yolact_state_dict(1234)
    rng = np.random.default_rng(5)
    x = fast_base_transform(rng.uniform(0, 255, (1, 200, 200, 3)).astype(np.float32))
    assert YolactConfig().conv_split_k == 0
    res = []
    for split in (0, 1):
        net = Yolact(sd, dataclasses.replace(YolactConfig(), conv_split_k=split), max_batch=1, input_size=200, fp16=True)
        net(x)
        res.append(net.fetch("proto", 1).copy())
        net.close()
    assert np.array_equal(res[0], res[1])
