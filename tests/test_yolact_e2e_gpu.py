"""Yolact end to end: HIP engine vs the numpy/C oracle model on the same seeded weights and images.
Every intermediate tensor and every output (indices, boxes, scores, coefficients, masks) must be bit-exact."""
import numpy as np
import pytest

from oracle.yolact_ref import YolactRef

pytestmark = pytest.mark.gpu


def _images(seed, n, size):
    from isegmi.yolact import fast_base_transform
    rng = np.random.default_rng(seed)
    return fast_base_transform(rng.uniform(0, 255, (n, size, size, 3)).astype(np.float32))


@pytest.fixture(scope="module")
def sd():
    from isegmi.weights import yolact_state_dict
    return yolact_state_dict(1234)


def _compare(net, ref, x, size, n):
    out = net(x)
    refd = ref.forward(x)
    for name in ("C3", "P3", "P5", "P7", "proto"):
        eng_name = {"C3": "res3.C"}.get(name, name)   # layer 1 = res3; a stage's final output has its own buffer
        got = net.fetch(eng_name, n)
        assert np.array_equal(got.reshape(ref.feats[name].shape), ref.feats[name]), name
    if net.fuse_heads:  # one (A*117)-wide conv: per pixel [Ax4 loc | Ax81 conf | Ax32 mask pre-tanh], A = 3 (9 for YOLACT++)
        from oracle import ora
        hc = net.fetch("headcat", n)
        A = net.cfg.num_priors
        heads = {"loc": hc[..., :4 * A].reshape(n, -1, 4), "conf": hc[..., 4 * A:85 * A].reshape(n, -1, 81),
                 "mask": ora.map_f32(np.ascontiguousarray(hc[..., 85 * A:]), 2).reshape(n, -1, 32)}
    else:
        heads = {k: net.fetch(k, n) for k in ("loc", "conf", "mask")}
    for name in ("loc", "conf", "mask"):
        assert np.array_equal(heads[name].reshape(ref.feats[name].shape), ref.feats[name]), name
    total = 0
    for i in range(n):
        r = refd[i]
        d = out[i]["detection"]
        if len(r["score"]) == 0:
            assert d is None
            continue
        for a, b in (("prior", "prior"), ("class", "cls"), ("score", "score"), ("box", "box"), ("mask", "mask")):
            assert np.array_equal(d[a], r[b]), a
        total += len(r["score"])
    return out, refd, total


def test_yolact_small_batch2_bit_exact(ffi, sd):
    from isegmi.yolact import Yolact, postprocess
    size = 200
    net = Yolact(sd, max_batch=2, input_size=size)
    ref = YolactRef(sd, max_size=550)
    x = _images(20261003, 2, size)
    out, refd, total = _compare(net, ref, x, size, 2)
    assert total > 0
    for i in range(2):
        cls, sc, boxes, masks = postprocess(out, 231, 187, batch_idx=i, score_threshold=0.15)
        rc, rs, rb, rm = YolactRef.postprocess(refd[i], 231, 187, 0.15)
        assert np.array_equal(cls, rc) and np.array_equal(sc, rs) and np.array_equal(boxes, rb) and np.array_equal(masks, rm)
    # determinism canary: same input twice -> identical bytes
    out2 = net(x)
    for i in range(2):
        for k in ("prior", "score", "box"):
            assert np.array_equal(out[i]["detection"][k], out2[i]["detection"][k])
    net.close()


def test_yolact_unfused_heads_bit_exact(ffi, sd):
    """The three separate prediction convs (the layout upstream uses) stay available and agree with the oracle too."""
    from isegmi.yolact import Yolact
    net = Yolact(sd, max_batch=1, input_size=200, fuse_heads=False)
    ref = YolactRef(sd, max_size=550)
    _, _, total = _compare(net, ref, _images(5, 1, 200), 200, 1)
    assert total > 0
    net.close()


def test_yolact_550_bit_exact(ffi, sd):
    from isegmi.yolact import Yolact, postprocess
    net = Yolact(sd, max_batch=1)
    ref = YolactRef(sd)
    x = _images(7, 1, 550)
    out, refd, total = _compare(net, ref, x, 550, 1)
    assert total >= 50
    cls, sc, boxes, masks = postprocess(out, 550, 550)
    rc, rs, rb, rm = YolactRef.postprocess(refd[0], 550, 550)
    assert np.array_equal(cls, rc) and np.array_equal(sc, rs) and np.array_equal(boxes, rb) and np.array_equal(masks, rm)
    assert masks.sum() > 0
    net.close()


def test_postprocess_after_second_forward_uses_the_new_batch(ffi, sd):
    """One net, two different batches, postprocess at the SAME output size after each (the `eval --images` loop): the masks and
    integer boxes of the second call must belong to the second batch; stale predictions are refused."""
    from isegmi.yolact import Yolact, postprocess
    size = 200
    net = Yolact(sd, max_batch=1, input_size=size)
    ref = YolactRef(sd, max_size=550)
    outs = []
    for seed in (31, 32):
        x = _images(seed, 1, size)
        out = net(x)
        r = ref.forward(x)[0]
        cls, sc, boxes, masks = postprocess(out, 190, 210)
        rc, rs, rb, rm = YolactRef.postprocess(r, 190, 210)
        assert len(rs) > 0
        assert np.array_equal(cls, rc) and np.array_equal(sc, rs) and np.array_equal(boxes, rb) and np.array_equal(masks, rm), seed
        outs.append((out, masks))
    assert outs[0][1].shape != outs[1][1].shape or not np.array_equal(outs[0][1], outs[1][1])
    with pytest.raises(RuntimeError, match="earlier forward"):
        postprocess(outs[0][0], 190, 210)
    net.close()


def test_yolact_550_bs8_bit_exact(ffi, sd):
    """BASELINE configs[1] at its own workload: eight 550x550 images in one batch, every detection field and the
    550x550 masks of every image against the oracle."""
    from isegmi.yolact import Yolact, postprocess
    net = Yolact(sd, max_batch=8)
    ref = YolactRef(sd)
    x = _images(20261003, 8, 550)
    out = net(x)
    refd = ref.forward(x)
    total = 0
    for i in range(8):
        r, d = refd[i], out[i]["detection"]
        assert d is not None and len(d["score"]) == len(r["score"])
        for a, b in (("prior", "prior"), ("class", "cls"), ("score", "score"), ("box", "box"), ("mask", "mask")):
            assert np.array_equal(d[a], r[b]), (i, a)
        cls, sc, boxes, masks = postprocess(out, 550, 550, batch_idx=i)
        rc, rs, rb, rm = YolactRef.postprocess(r, 550, 550)
        assert np.array_equal(cls, rc) and np.array_equal(sc, rs) and np.array_equal(boxes, rb) and np.array_equal(masks, rm), i
        total += len(rs)
    assert total >= 400
    net.close()


def test_back_to_back_forwards_are_independent(ffi, sd):
    """Cross-step overlap (tail stream): forwards queued without a host sync must not disturb each other.
    A, B alternate five times with no sync in between; the final results must equal a clean run of the last batch."""
    from isegmi.yolact import Yolact
    size = 200
    net = Yolact(sd, max_batch=2, input_size=size)
    xa, xb = _images(11, 2, size), _images(12, 2, size)
    clean = {}
    for name, x in (("a", xa), ("b", xb)):
        net.upload(x); net.forward_device(2); net.postprocess_device(150, 170); net.sync()
        clean[name] = {k: net.fetch(k, 2) for k in ("det.count", "det.score", "det.prior", "det.box", "det.masks", "det.box_int")}
    da = ffi.DeviceBuffer.from_numpy(xa); db = ffi.DeviceBuffer.from_numpy(xb)
    import ctypes as C
    for i in range(5):
        for name, d in (("a", da), ("b", db)):
            ffi.check(ffi.lib().isegmi_yolact_forward(net._h, d.ptr, 2))
            net.postprocess_device(150, 170)
    net.sync()
    for k, v in clean["b"].items():
        got = net.fetch(k, 2)
        if k == "det.masks":
            cnt = clean["b"]["det.count"]
            assert all(np.array_equal(got[i, : cnt[i]], v[i, : cnt[i]]) for i in range(2))
        else:
            assert np.array_equal(got, v), k
    assert not np.array_equal(clean["a"]["det.score"], clean["b"]["det.score"])
    net.close()


def test_yolact_base_and_im700_configs_bit_exact(ffi):
    """yolact_base (ResNet101-FPN) and yolact_im700 (R101, 700 px, scales int(s/550*700)) on the full 700x700
    image (the ResNet101 backbone at a small input is covered by the YOLACT++ and Mask R-CNN R101 tests)."""
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, YolactConfig, postprocess
    sd101 = yolact_state_dict(1234, depth=101)
    cfg = YolactConfig.im700()
    assert cfg.pred_scales == (30, 61, 122, 244, 488) and cfg.max_size == 700 and cfg.depth == 101
    ref = YolactRef(sd101, max_size=700, scales=cfg.pred_scales, depth=101)
    net = Yolact(sd101, cfg, max_batch=1)
    assert net.size == 700 and net.priors.shape == (30963, 4)
    x = _images(4, 1, 700)
    out, refd, total = _compare(net, ref, x, 700, 1)
    cls, sc, boxes, masks = postprocess(out, 640, 480)
    rc, rs, rb, rm = YolactRef.postprocess(refd[0], 640, 480)
    assert np.array_equal(cls, rc) and np.array_equal(sc, rs) and np.array_equal(boxes, rb) and np.array_equal(masks, rm)
    net.close()


def test_yolact_darknet53_bit_exact(ffi):
    """yolact_darknet53_config (the Darknet53-FPN row of README.md:209-214): DarkNetBackbone([1, 2, 8, 8, 4]) with LeakyReLU(0.1)
    and the shortcut added after the activation, layers 2-4 into the usual FPN / protonet / heads."""
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, YolactConfig, darknet_base_transform, postprocess
    cfg = YolactConfig.darknet53()
    sd = yolact_state_dict(31, backbone="darknet53")
    assert sd["fpn.lat_layers.0.weight"].shape == (256, 1024, 1, 1) and sd["backbone.layers.4.4.conv2.weight"].shape == (1024, 512, 3, 3)
    size = 200
    rng = np.random.default_rng(8)
    x = darknet_base_transform(rng.uniform(0, 255, (2, size, size, 3)).astype(np.float32))
    assert float(x.max()) <= 1.0 and float(x.min()) >= 0.0
    net = Yolact(sd, cfg, max_batch=2, input_size=size)
    ref = YolactRef(sd, max_size=550)
    out = net(x)
    refd = ref.forward(x)
    for name, eng in (("C3", "backbone.layers.2.8.out"), ("P3", "P3"), ("P7", "P7"), ("proto", "proto")):
        assert np.array_equal(net.fetch(eng, 2).reshape(ref.feats[name].shape), ref.feats[name]), name
    total = 0
    for i in range(2):
        r, d = refd[i], out[i]["detection"]
        if len(r["score"]) == 0:
            assert d is None
            continue
        for a, b in (("prior", "prior"), ("class", "cls"), ("score", "score"), ("box", "box"), ("mask", "mask")):
            assert np.array_equal(d[a], r[b]), a
        total += len(r["score"])
        cls, sc, boxes, masks = postprocess(out, size, size, batch_idx=i)
        rc, rs, rb, rm = YolactRef.postprocess(r, size, size)
        assert np.array_equal(cls, rc) and np.array_equal(sc, rs) and np.array_equal(boxes, rb) and np.array_equal(masks, rm)
    assert total > 0
    net.close()


def test_yolact_plus_dcn_backbones_bit_exact(ffi):
    """SURVEY 8f rank 4, the YOLACT++ rows of README.md:216-221: DCNv2 3x3s in the backbone (every block of layers 2-4 on
    ResNet50, every third on ResNet101), nine rectangular anchors per cell (three scales x three ratios)."""
    from isegmi.weights import dcn_blocks, yolact_state_dict
    from isegmi.yolact import Yolact, YolactConfig, postprocess
    assert sorted(dcn_blocks(101, (0, 4, 23, 3), 3)) == [(1, 0), (1, 3)] + [(2, b) for b in range(0, 23, 3)] + [(3, 0)]
    assert len(dcn_blocks(50, (0, 4, 6, 3), 1)) == 13
    for cfg, size in ((YolactConfig.plus_resnet50(), 200), (YolactConfig.plus_base(), 136)):
        sd = yolact_state_dict(77, depth=cfg.depth, num_priors=9, dcn_layers=cfg.dcn_layers, dcn_interval=cfg.dcn_interval)
        nb = 2 if cfg.depth == 50 else 1  # (oracle time: the ResNet101 pass runs one image)
        net = Yolact(sd, cfg, max_batch=nb, input_size=size)
        assert net.priors.shape[1] == 4 and net.priors.shape[0] % 9 == 0
        ref = YolactRef(sd, max_size=550, depth=cfg.depth, scales_per_level=3, square=False)
        x = _images(5 + size, nb, size)
        out, refd, total = _compare(net, ref, x, size, nb)
        assert np.array_equal(net.priors, ref.feats["priors"])
        assert total > 0
        # one deformable block in detail: offsets / mask logits, sampled columns
        om = net.fetch("backbone.layers.1.0.om", nb)
        assert om.shape[-1] == 27 and float(np.abs(om[..., :18]).mean()) > 0.05  # the synthetic offsets are not all ~0
        cls, sc, boxes, masks = postprocess(out, size, size)
        rc, rs, rb, rm = YolactRef.postprocess(refd[0], size, size)
        assert np.array_equal(cls, rc) and np.array_equal(sc, rs) and np.array_equal(boxes, rb) and np.array_equal(masks, rm)
        net.close()


def test_yolact_plus_mask_rescoring_bit_exact(ffi):
    """YOLACT++ fast mask re-scoring (use_maskiou / rescore_mask): postprocess returns scores = [box scores, box scores x
    mask IoU], the mask IoU from FastMaskIoUNet on the cropped proto-resolution masks.  256 px is the smallest input whose
    64x64 prototypes survive the net's five unpadded stride-2 3x3 convs."""
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, YolactConfig, postprocess
    cfg = YolactConfig.plus_resnet50()
    size = 256
    sd = yolact_state_dict(77, depth=50, num_priors=9, dcn_layers=cfg.dcn_layers, dcn_interval=cfg.dcn_interval, maskiou=True)
    net = Yolact(sd, cfg, max_batch=2, input_size=size)
    ref = YolactRef(sd, max_size=550, depth=50, scales_per_level=3, square=False)
    x = _images(91, 2, size)
    out, refd, total = _compare(net, ref, x, size, 2)
    assert total > 0
    for i in range(2):
        for thr in (0.0, 0.3):
            cls, sc, boxes, masks = postprocess(out, size, size, batch_idx=i, score_threshold=thr)
            rc, rs, rb, rm = YolactRef.postprocess(refd[i], size, size, thr)
            assert isinstance(sc, list) and len(sc) == 2
            assert np.array_equal(cls, rc) and np.array_equal(sc[0], rs) and np.array_equal(boxes, rb) and np.array_equal(masks, rm)
            rms = ref.mask_scores(refd[i], thr)
            assert np.array_equal(sc[1], rms), (i, thr)
            if len(rms):
                assert (rms >= 0).all() and float(rms.max()) > 0.0  # the synthetic net gives non-trivial IoU predictions
    net.close()


def test_yolact_smooth_images_bit_exact(ffi, sd):
    """Second input suite: smooth low-frequency fields -> clustered boxes, heavy fast-NMS suppression."""
    from conftest import smooth_field
    from isegmi.yolact import Yolact, fast_base_transform
    size = 200
    x = fast_base_transform(np.stack([smooth_field(31 + i, size, size) for i in range(2)]))
    net = Yolact(sd, max_batch=2, input_size=size)
    _compare(net, YolactRef(sd, max_size=550), x, size, 2)
    net.close()


def test_yolact_pipelined_heads_back_to_back(ffi, sd):
    """Cross-step pipelining ("pipeline_heads", default on): FPN + protonet + heads of forward i run on their own streams while
    forward i+1's backbone is already running.  Several forwards enqueued back to back without a sync, on CHANGING inputs,
    must leave exactly the results of the last input computed alone (fences: C3-C5 vs the lateral convs, head buffers vs the
    previous Detect / postprocess), and switching the mode off without a sync in between must be safe too."""
    from isegmi.yolact import Yolact
    size = 200
    keys = ("det.count", "det.score", "det.prior", "det.box", "det.class", "det.masks", "proto")
    xs = [_images(300 + i, 2, size) for i in range(4)]
    net = Yolact(sd, max_batch=2, input_size=size)
    net.set_param("pipeline_heads", 0.0)
    net.upload(xs[3]); net.forward_device(2); net.postprocess_device(size, size); net.sync()
    want = {k: net.fetch(k, 2) for k in keys}
    assert int(want["det.count"].sum()) > 0
    net.set_param("pipeline_heads", 1.0)
    for rep in range(2):
        for x in xs:
            net.upload(x); net.forward_device(2); net.postprocess_device(size, size)
        net.sync()
        for k in keys:
            assert np.array_equal(net.fetch(k, 2), want[k]), (rep, k)
    # pipelined forwards on other inputs, then WITHOUT a sync a non-pipelined one on the reference input
    for x in xs[:3]:
        net.upload(x); net.forward_device(2); net.postprocess_device(size, size)
    net.set_param("pipeline_heads", 0.0)
    net.upload(xs[3]); net.forward_device(2); net.postprocess_device(size, size); net.sync()
    for k in keys:
        assert np.array_equal(net.fetch(k, 2), want[k]), ("switch", k)
    net.close()


def test_yolact_hipgraph_replay_matches_eager(ffi, sd):
    """"graph" param: the forward is captured into a hipGraph on its second call and replayed afterwards; results must
    equal the eager multi-stream path bit for bit, also after the input buffer's CONTENT changes (same pointer) and
    after a parameter change (which must drop the captured graph)."""
    from isegmi.yolact import Yolact
    size = 200
    net = Yolact(sd, max_batch=2, input_size=size)
    keys = ("det.count", "det.score", "det.prior", "det.box", "det.class", "det.masks")
    def run(x):
        net.upload(x); net.forward_device(2); net.postprocess_device(150, 170); net.sync()
        return {k: net.fetch(k, 2) for k in keys}
    xa, xb = _images(21, 2, size), _images(22, 2, size)
    ea, eb = run(xa), run(xb)
    net.set_param("graph", 1.0)
    for rep in range(3):  # eager warm-up, capture, replay
        for name, x, want in (("a", xa, ea), ("b", xb, eb)):
            got = run(x)
            for k in keys:
                if k == "det.masks":
                    assert all(np.array_equal(got[k][i, : want["det.count"][i]], want[k][i, : want["det.count"][i]]) for i in range(2))
                else:
                    assert np.array_equal(got[k], want[k]), (rep, name, k)
    import ctypes as C
    cap, rep_, fail = C.c_int64(), C.c_int64(), C.c_int64()
    ffi.check(ffi.lib().isegmi_engine_graph_stats(net._h, C.byref(cap), C.byref(rep_), C.byref(fail)))
    assert cap.value == 1 and rep_.value >= 4 and fail.value == 0
    net.set_param("nms_conf_thresh", 0.12)
    g3 = run(xa)
    net.set_param("graph", 0.0)
    e3 = run(xa)
    assert np.array_equal(g3["det.count"], e3["det.count"]) and np.array_equal(g3["det.score"], e3["det.score"])
    assert not np.array_equal(e3["det.score"], ea["det.score"])
    net.close()


def _iou_norm(a, b):
    ax1, ay1, ax2, ay2 = [a[:, i][:, None] for i in range(4)]
    bx1, by1, bx2, by2 = [b[:, i][None, :] for i in range(4)]
    iw = np.clip(np.minimum(ax2, bx2) - np.maximum(ax1, bx1), 0, None); ih = np.clip(np.minimum(ay2, by2) - np.maximum(ay1, by1), 0, None)
    inter = iw * ih
    return inter / ((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - inter + 1e-12)


def test_yolact_fp16_mode_close_to_fp16_oracle(ffi, sd):
    """Optional fp16-storage / f16-MFMA mode of the Yolact engine.  TOLERANCE (stated): the 16-term sum inside one f16 MFMA is
    unordered, so features are compared at 5e-3 of the tensor's max magnitude against an oracle that rounds the same tensors to
    fp16; detections are matched by IoU because near-tied scores may swap: >= 90 % of the oracle's detections must have a
    same-class partner with IoU >= 0.9 and |score diff| <= 0.01; masks of matched pairs must agree on >= 99 % of the pixels."""
    from isegmi.yolact import Yolact, postprocess
    size = 200
    net = Yolact(sd, max_batch=2, input_size=size, fp16=True)
    ref = YolactRef(sd, max_size=550, fp16=True)
    x = _images(20261003, 2, size)
    out = net(x)
    refd = ref.forward(x)
    for name in ("P3", "P5", "P7"):
        g = net.fetch(name, 2)
        assert g.dtype == np.float16
        r = ref.feats[name]
        assert np.abs(g.astype(np.float32).reshape(r.shape) - r).max() <= 5e-3 * np.abs(r).max(), name
    g = net.fetch("proto", 2); r = ref.feats["proto"]
    assert g.dtype == np.float32 and np.abs(g.reshape(r.shape) - r).max() <= 5e-3 * np.abs(r).max()
    hc = net.fetch("headcat", 2)
    assert hc.dtype == np.float32
    assert np.abs(hc[..., 12:255].reshape(ref.feats["conf"].shape) - ref.feats["conf"]).max() <= 5e-3 * np.abs(ref.feats["conf"]).max()
    total = 0
    for i in range(2):
        r, d = refd[i], out[i]["detection"]
        assert d is not None and abs(len(d["score"]) - len(r["score"])) <= 5
        iou = _iou_norm(r["box"], d["box"])
        same = r["cls"][:, None] == d["class"][None, :]
        close = np.abs(r["score"][:, None] - d["score"][None, :]) <= 0.01
        ok = (iou >= 0.9) & same & close
        matched = np.any(ok, axis=1)
        assert matched.mean() >= 0.9, matched.mean()
        total += len(r["score"])
        cls, sc, boxes, masks = postprocess(out, 231, 187, batch_idx=i, score_threshold=0.0)
        rc, rs, rb, rm = YolactRef.postprocess(r, 231, 187, 0.0)
        j = np.argmax(np.where(ok, iou, -1), axis=1)
        agree = (masks[j[matched]] == rm[matched]).mean()
        assert agree >= 0.99, agree
    assert total > 20
    net.close()
