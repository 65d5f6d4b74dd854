"""The launch forms the defaults no longer take: per-layer launches instead of grouped convolutions (`conv_groups` 0), the per-level RPN
selection on side streams instead of the (level, image)-batched one (`rpn_select_groups` 0; also what PRE_NMS_TOP_N_TEST outside 257..1024 falls back
to), the two forced groupings (1 / 2), and the stream joins that go with them.  They stay selectable for A/B runs and for configurations the grouped
forms do not cover, so they are tested: every tensor the heads produce must be BIT-IDENTICAL to the default path's, fp32 and fp16, one image and two."""
import dataclasses

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    from isegmi.weights import maskrcnn_state_dict
    return maskrcnn_state_dict(1234)


def _maskrcnn_run(sd, x, hw, fp16, params, cfg=None):
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig
    n = x.shape[0]
    model = MaskRCNN(sd, x.shape[1], x.shape[2], cfg=cfg or MaskRCNNConfig(), max_batch=n, fp16=fp16)
    for k, v in params.items():
        model.set_param(k, float(v))
    res = []
    for rep in range(2):   # twice: the second forward runs against the first one's tail (WAR fences, side-stream joins)
        model(x, hw)
        model.paste_device(x.shape[1], x.shape[2]); model.sync()
        res.append({k: model.fetch(k, n) for k in ("proposal_count", "proposals", "proposal_scores", "det.count", "det.box", "det.score", "det.label",
                                                   "det.mask28", "det.masks")})
    model.close()
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), ("second forward differs", k)
    return res[0]


def _trim(r):
    """only what is defined: rows below the counts"""
    out = {}
    pc, dc = r["proposal_count"], r["det.count"]
    out["pc"], out["dc"] = pc, dc
    for k in ("proposals", "proposal_scores"):
        out[k] = [r[k][n, : pc[n]] for n in range(len(pc))]
    for k in ("det.box", "det.score", "det.label", "det.mask28", "det.masks"):
        out[k] = [r[k][n, : dc[n]] for n in range(len(dc))]
    return out


def _same(a, b):
    a, b = _trim(a), _trim(b)
    for k in a:
        if k in ("pc", "dc"):
            assert np.array_equal(a[k], b[k]), k
        else:
            for n in range(len(a[k])):
                assert np.array_equal(a[k][n], b[k][n]), (k, n)


@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("N", [1, 2])
def test_maskrcnn_fallback_launch_forms_are_bit_identical_to_the_default(ffi, sd, fp16, N):
    from isegmi.maskrcnn import prepare_images
    rng = np.random.default_rng(20261003 + N)
    imgs = [rng.uniform(0, 255, s + (3,)).astype(np.float32) for s in [(250, 340), (256, 300)][:N]]
    x, hw = prepare_images(imgs)
    base = _maskrcnn_run(sd, x, hw, fp16, {})
    assert int(base["det.count"].sum()) > 10 * N
    variants = [dict(rpn_select_groups=0), dict(rpn_select_groups=1), dict(rpn_select_groups=2), dict(rpn_select_groups=0, rpn_select_on_tail=0),
                dict(rpn_select_groups=0, rpn_select_on_tail=1), dict(multi_stream=0), dict(box_nms_chip_wide=0)]
    if not fp16:   # the fp16 engine has no grouped convolutions to switch off
        variants += [dict(conv_groups=0), dict(conv_groups=0, rpn_select_groups=0), dict(conv_groups=0, rpn_select_groups=2), dict(conv_groups=1, rpn_select_groups=0)]
    for v in variants:
        try:
            _same(_maskrcnn_run(sd, x, hw, fp16, v), base)
        except AssertionError as e:
            raise AssertionError("variant %r: %s" % (v, e))


@pytest.mark.parametrize("fp16", [False, True])
def test_maskrcnn_pre_nms_2000_takes_the_per_level_path_whatever_the_grouping_says(ffi, sd, fp16):
    """PRE_NMS_TOP_N_TEST = 2000 is outside the batched selection's 257..1024: the engine falls back to the per-level launches (single-block 6144-box NMS)
    with any `rpn_select_groups`; grouped and per-layer convolutions give the same bits."""
    from isegmi.maskrcnn import MaskRCNNConfig, prepare_images
    rng = np.random.default_rng(77)
    x, hw = prepare_images([rng.uniform(0, 255, (250, 340, 3)).astype(np.float32), rng.uniform(0, 255, (256, 300, 3)).astype(np.float32)])
    cfg = dataclasses.replace(MaskRCNNConfig(), RPN_PRE_NMS_TOP_N_TEST=2000, RPN_POST_NMS_TOP_N_TEST=2000)
    base = _maskrcnn_run(sd, x, hw, fp16, {}, cfg)
    assert (base["proposal_count"] == 1000).all()
    for v in (dict(rpn_select_groups=1), dict(rpn_select_groups=2), dict(conv_groups=0), dict(rpn_select_on_tail=0)):
        if fp16 and "conv_groups" in v:
            continue
        _same(_maskrcnn_run(sd, x, hw, fp16, v, cfg), base)


@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("N", [1, 2])
def test_yolact_fallback_launch_forms_are_bit_identical_to_the_default(ffi, fp16, N):
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, fast_base_transform
    sdy = yolact_state_dict(1234)
    rng = np.random.default_rng(3 + N)
    x = fast_base_transform(rng.uniform(0, 255, (N, 200, 200, 3)).astype(np.float32))
    keys = ("det.count", "det.score", "det.box", "det.class", "det.coeff", "proto")

    def run(params, fuse_heads=True):
        net = Yolact(sdy, max_batch=N, input_size=200, fp16=fp16, fuse_heads=fuse_heads)
        for k, v in params.items():
            net.set_param(k, float(v))
        outs = []
        for rep in range(2):
            net(x)
            net.postprocess_device(200, 200); net.sync()
            r = {k: net.fetch(k, N) for k in keys}
            r["masks"] = net.fetch("det.masks", N)
            outs.append(r)
        net.close()
        for k in outs[0]:
            if k != "masks":
                assert np.array_equal(outs[0][k], outs[1][k]), ("second forward differs", k)
        return outs[0]

    def same(a, b):
        assert np.array_equal(a["det.count"], b["det.count"]) and np.array_equal(a["proto"], b["proto"])
        for n in range(N):
            c = int(a["det.count"][n])
            for k in ("det.score", "det.box", "det.class", "det.coeff", "masks"):
                assert np.array_equal(a[k][n, :c], b[k][n, :c]), (k, n)

    base = run({})
    assert int(base["det.count"].sum()) > 0
    variants = [dict(multi_stream=0)]
    if not fp16:
        variants += [dict(conv_groups=0), dict(conv_groups=0, multi_stream=0)]
    for v in variants:
        try:
            same(run(v), base)
        except AssertionError as e:
            raise AssertionError("variant %r: %s" % (v, e))
    if not fp16:   # (the fp16 engine has the fused prediction head only)
        same(run({}, fuse_heads=False), base)
