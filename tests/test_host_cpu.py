"""Host-side logic (no GPU): priors, anchors, image preparation, BoxList, record packing, batch sharding,
and a world_size-2 gloo run of the multi-rank detection gather."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_yolact_priors_match_oracle_and_count():
    from isegmi.yolact import YolactConfig, _level_sizes, make_priors
    from oracle.yolact_ref import make_priors as ref_priors
    cfg = YolactConfig()
    sizes = _level_sizes(550)
    assert sizes == [69, 35, 18, 9, 5]
    tot = 0
    for s, sc in zip(sizes, cfg.pred_scales):
        a = make_priors(s, s, sc, 550, cfg.pred_aspect_ratios); b = ref_priors(s, s, sc, 550)
        assert np.array_equal(a, b)
        tot += len(a)
    assert tot == 19248


def test_yolact_plus_config_priors_dcn_blocks_and_weights():
    """YOLACT++ host logic: nine rectangular anchors per cell in scale-major / ratio-minor order (57 744 priors at 550 px), the
    upstream _make_layer rule for which bottlenecks carry a DCNv2 3x3, and the synthetic state dict's upstream names / shapes."""
    from isegmi.weights import dcn_blocks, yolact_state_dict
    from isegmi.yolact import YolactConfig, _level_sizes, make_priors
    from oracle.yolact_ref import make_priors as ref_priors
    cfg = YolactConfig.plus_base()
    assert cfg.num_priors == 9 and not cfg.use_square_anchors and cfg.use_maskiou and cfg.depth == 101
    tot = 0
    for l, s in enumerate(_level_sizes(550)):
        sc = cfg.level_scales(l)
        assert len(sc) == 3 and sc[0] == cfg.pred_scales[l] and abs(sc[2] / sc[0] - 2 ** (2 / 3.0)) < 1e-12
        a = make_priors(s, s, sc, cfg.max_size, cfg.pred_aspect_ratios, square=False)
        b = ref_priors(s, s, sc, cfg.max_size, square=False)
        assert np.array_equal(a, b)
        tot += len(a)
    assert tot == 57744
    first = make_priors(2, 2, (24.0, 30.0), 550, (1.0, 0.5, 2.0), square=False)[:6]
    assert np.allclose(first[:, 2] / first[:, 3], [1.0, 0.5, 2.0, 1.0, 0.5, 2.0])  # w/h = aspect ratio, ratios inside scales
    assert np.allclose(first[[0, 3], 2] * 550, [24.0, 30.0])
    # resnet101_dcn_inter3_backbone: first block of layers 2-4 and every third after it; resnet50_dcnv2_backbone: all of layers 2-4
    assert sorted(dcn_blocks(101, (0, 4, 23, 3), 3)) == [(1, 0), (1, 3)] + [(2, b) for b in range(0, 23, 3)] + [(3, 0)]
    assert sorted(dcn_blocks(50, (0, 4, 6, 3), 1)) == [(1, b) for b in range(4)] + [(2, b) for b in range(6)] + [(3, b) for b in range(3)]
    assert dcn_blocks(50, (0, 0, 0, 0), 1) == set()
    c50 = YolactConfig.plus_resnet50()
    sd = yolact_state_dict(5, c50.depth, c50.num_priors, c50.dcn_layers, c50.dcn_interval, True)
    assert sd["backbone.layers.1.0.conv2.conv_offset_mask.weight"].shape == (27, 128, 3, 3)
    assert sd["backbone.layers.3.2.conv2.bias"].shape == (512,) and "backbone.layers.0.0.conv2.bias" not in sd
    assert sd["prediction_layers.0.conf_layer.weight"].shape == (9 * 81, 256, 3, 3)
    assert [sd["maskiou_net.%d.weight" % i].shape[:2] for i in (0, 2, 4, 6, 8, 10)] == [(8, 1), (16, 8), (32, 16), (64, 32), (128, 64), (80, 128)]


def test_yolact_darknet53_config_weights_and_transform():
    """yolact_darknet53_config host logic: upstream DarkNetBackbone state-dict names / shapes, the 256/512/1024-channel FPN inputs,
    darknet_transform (x / 255, BGR -> RGB) and the oracle's backbone on a tiny input (strides 8 / 16 / 32)."""
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import YolactConfig, darknet_base_transform
    from oracle.yolact_ref import YolactRef
    cfg = YolactConfig.darknet53()
    assert cfg.backbone == "darknet53" and cfg.num_priors == 3 and cfg.use_square_anchors
    sd = yolact_state_dict(3, backbone="darknet53")
    assert sd["backbone._preconv.0.weight"].shape == (32, 3, 3, 3) and sd["backbone.layers.0.0.0.weight"].shape == (64, 32, 3, 3)
    assert sd["backbone.layers.2.8.conv1.weight"].shape == (128, 256, 1, 1) and sd["backbone.layers.2.8.conv2.weight"].shape == (256, 128, 3, 3)
    assert "backbone.layers.2.9.conv1.weight" not in sd and "backbone.conv1.weight" not in sd
    assert [sd["fpn.lat_layers.%d.weight" % i].shape[1] for i in range(3)] == [1024, 512, 256]
    n_blocks = sum(1 for k in sd if k.endswith(".conv2.weight") and k.startswith("backbone.layers."))
    assert n_blocks == 1 + 2 + 8 + 8 + 4
    img = np.zeros((1, 2, 2, 3), np.float32); img[0, 0, 0] = [255.0, 51.0, 0.0]
    t = darknet_base_transform(img)
    assert np.array_equal(t[0, 0, 0], np.float32([0.0, np.float32(51.0) / np.float32(255.0), 1.0]))
    ref = YolactRef(sd)
    c3, c4, c5 = ref._darknet(np.random.default_rng(0).uniform(0, 1, (1, 64, 64, 3)).astype(np.float32))
    assert c3.shape == (1, 8, 8, 256) and c4.shape == (1, 4, 4, 512) and c5.shape == (1, 2, 2, 1024)


def test_maskrcnn_anchors_and_shapes():
    from isegmi.maskrcnn import generate_anchors, grid_anchors, level_shapes
    from oracle.maskrcnn_ref import cell_anchors, grid_anchors as ref_grid
    assert level_shapes(800, 1344) == [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    assert sum(h * w * 3 for h, w in level_shapes(800, 1344)) == 268569
    for stride, size in zip((4, 8, 16, 32, 64), (32, 64, 128, 256, 512)):
        a = generate_anchors(stride, size, (0.5, 1.0, 2.0))
        assert np.array_equal(a, cell_anchors(stride, size))
        assert np.array_equal(grid_anchors(3, 4, stride, a), ref_grid(3, 4, stride, a))
    assert np.array_equal(generate_anchors(16, 128, (0.5, 1.0, 2.0))[1], [-56, -56, 71, 71])


def test_prepare_images_and_boxlist():
    from isegmi.maskrcnn import PIXEL_MEAN, BoxList, prepare_images
    ims = [np.full((30, 50, 3), 128, np.float32), np.full((40, 33, 3), 10, np.float32)]
    x, hw = prepare_images(ims)
    assert x.shape == (2, 64, 64, 3) and hw.tolist() == [[30, 50], [40, 33]]
    assert np.allclose(x[0, 0, 0], 128 - np.asarray(PIXEL_MEAN, np.float32)) and not x[0, 30:].any() and not x[1, :, 33:].any()
    bl = BoxList(np.array([[10, 20, 30, 40]], np.float32), (100, 50))
    bl.add_field("scores", np.array([0.9], np.float32))
    r = bl.resize((200, 150))
    assert np.allclose(r.bbox, [[20, 60, 60, 120]]) and r.size == (200, 150) and r.get_field("scores")[0] == np.float32(0.9)
    assert len(bl[np.array([False])]) == 0


def test_bn_folding_matches_oracle_formulae():
    from isegmi.weights import fold_batchnorm, fold_frozen_batchnorm, yolact_state_dict
    from oracle.maskrcnn_ref import _frozen_bn
    from oracle.yolact_ref import _fold_bn
    sd = yolact_state_dict(7)
    a = fold_batchnorm(sd, "backbone.bn1"); b = _fold_bn(sd, "backbone.bn1")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a = fold_frozen_batchnorm(sd, "backbone.bn1"); b = _frozen_bn(sd, "backbone.bn1")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_record_pack_roundtrip_and_sharding():
    from isegmi.dist import pack_records, record_bytes, shard_batch, unpack_records
    rng = np.random.default_rng(0)
    n, K = 3, 100
    rec = dict(count=rng.integers(0, 101, n).astype(np.int32), box=rng.standard_normal((n, K, 4)).astype(np.float32),
               score=rng.uniform(0, 1, (n, K)).astype(np.float32), cls=rng.integers(0, 80, (n, K)).astype(np.int32),
               coeff=rng.standard_normal((n, K, 32)).astype(np.float32))
    buf = pack_records(rec["count"], rec["box"], rec["score"], rec["cls"], rec["coeff"])
    assert buf.nbytes == record_bytes(n)
    back = unpack_records(buf, n)
    assert all(np.array_equal(back[k], rec[k]) for k in rec)
    for world in (1, 2, 3, 8):
        spans = [shard_batch(16, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 16 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    assert [shard_batch(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]


WORKER = r'''
import os, sys
sys.path[:0] = [sys.argv[1], os.path.join(sys.argv[1], "instancesegmentation-jittor_amd")]
import numpy as np, torch, torch.distributed as dist
from isegmi.dist import pack_records, unpack_records, shard_batch, gather_records
from oracle import ora
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
GLOBAL, K, P = 6, 100, 400
def detect_image(i):   # the oracle stands in for the device path: same record layout
    rng = np.random.default_rng(1000 + i)
    conf = rng.standard_normal((P, 81)).astype(np.float32); conf[:, 0] += 3; conf[rng.integers(0, P, 40), rng.integers(1, 81, 40)] += 7
    pri = np.concatenate([rng.uniform(0.1, 0.9, (P, 2)), rng.uniform(0.05, 0.4, (P, 2))], 1).astype(np.float32)
    loc = (rng.standard_normal((P, 4)) * 0.5).astype(np.float32); msk = np.tanh(rng.standard_normal((P, 32))).astype(np.float32)
    return ora.yolact_detect(ora.softmax(conf), ora.yolact_decode(loc, pri), msk)
def records(lo, hi):
    n = hi - lo
    cnt = np.zeros(n, np.int32); box = np.zeros((n, K, 4), np.float32); sc = np.zeros((n, K), np.float32)
    cl = np.full((n, K), -1, np.int32); co = np.zeros((n, K, 32), np.float32)
    for j, i in enumerate(range(lo, hi)):
        d = detect_image(i); c = len(d["score"]); cnt[j] = c
        box[j, :c] = d["box"]; sc[j, :c] = d["score"]; cl[j, :c] = d["cls"]; co[j, :c] = d["mask"]
    return pack_records(cnt, box, sc, cl, co)
lo, hi = shard_batch(GLOBAL, rank, world)
mine = records(lo, hi)
def allgather(buf):
    t = torch.from_numpy(buf.copy()); outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return [o.numpy() for o in outs]
parts = gather_records(mine, world, allgather)
full = unpack_records(records(0, GLOBAL), GLOBAL)
off = 0
for r in range(world):
    a, b = shard_batch(GLOBAL, r, world)
    got = unpack_records(parts[r], b - a)
    for k in got:
        assert np.array_equal(got[k], full[k][a:b]), (rank, r, k)
    off += b - a
assert off == GLOBAL
# Mask R-CNN records (28x28 masks of the FPN head, 14x14 of MaskRCNNC4Predictor), two consecutive steps with different
# content and no barrier in between: every rank must see each step's records of every other rank, never a mix
from isegmi.dist import pack_maskrcnn_records, unpack_maskrcnn_records, maskrcnn_record_bytes
def mrec(lo, hi, M, step):
    n = hi - lo
    cnt = np.zeros(n, np.int32); box = np.zeros((n, K, 4), np.float32); sc = np.zeros((n, K), np.float32)
    lab = np.zeros((n, K), np.int32); mk = np.zeros((n, K, M, M), np.float32)
    for j, i in enumerate(range(lo, hi)):
        rng = np.random.default_rng(77 * step + i)
        c = int(rng.integers(0, K + 1)); cnt[j] = c
        box[j, :c] = rng.uniform(0, 800, (c, 4)); sc[j, :c] = np.sort(rng.uniform(0, 1, c))[::-1]
        lab[j, :c] = rng.integers(1, 81, c); mk[j, :c] = rng.uniform(0, 1, (c, M, M))
    return pack_maskrcnn_records(cnt, box, sc, lab, mk)
for M in (28, 14):
    gathered = []
    for step in range(2):
        mine = mrec(lo, hi, M, step)
        assert mine.nbytes == maskrcnn_record_bytes(hi - lo, K, M)
        gathered.append(gather_records(mine, world, allgather))
    for step in range(2):
        full = unpack_maskrcnn_records(mrec(0, GLOBAL, M, step), GLOBAL, K, M)
        for r in range(world):
            a, b = shard_batch(GLOBAL, r, world)
            got = unpack_maskrcnn_records(gathered[step][r], b - a, K, M)
            for k in got:
                assert np.array_equal(got[k], full[k][a:b]), (rank, r, k, M, step)
dist.barrier(); dist.destroy_process_group()
open(os.path.join(sys.argv[2], "ok_%d" % rank), "w").write("ok")
'''


def test_world2_gloo_detection_gather(tmp_path):
    pytest.importorskip("torch")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    import socket
    with socket.socket() as sk:   # a free port, so back-to-back runs never collide on TIME_WAIT
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT, str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_cocodemo_holds_exactly_one_engine(monkeypatch):
    """COCODemo builds ONE engine at the largest canvas the resize rule can produce (either orientation) and keeps it for every image;
    only a larger batch size than it was built for replaces it."""
    from isegmi import predictor

    class Fake:
        alive = 0

        def __init__(self, sd, H, W, cfg=None, max_batch=1, **kw):
            self.key, self.max_batch = (H, W), max_batch; Fake.alive += 1

        def reserve(self):
            return (1, 2)

        def close(self):
            Fake.alive -= 1
    monkeypatch.setattr(predictor, "MaskRCNN", Fake)
    demo = predictor.COCODemo(state_dict={"x": 0}, max_batch=2)
    e = demo.engine()
    assert e.key == (1344, 1344) and e.max_batch == 2 and demo.memory == (1, 2)
    assert demo.engine() is e and demo.engine(1) is e and Fake.alive == 1
    e4 = demo.engine(4)
    assert e4 is not e and e4.max_batch == 4 and Fake.alive == 1
    demo.close()
    assert Fake.alive == 0
    small = predictor.COCODemo(state_dict={"x": 0}, min_image_size=192, max_image_size=320)
    assert small.engine().key == (320, 320)
    small.close()


def test_overlay_draws_mask_box_and_class_label():
    """run_on_opencv_image's overlay (README.md:331-334): mask tint, box outline and the "class: score" text at the box corner"""
    from isegmi.maskrcnn import BoxList
    from isegmi.predictor import COCODemo
    demo = COCODemo.__new__(COCODemo)
    demo.confidence_threshold = 0.5
    image = np.zeros((100, 160, 3), np.uint8)
    bl = BoxList([[10, 20, 80, 90]], (160, 100))
    bl.add_field("scores", np.array([0.87], np.float32)); bl.add_field("labels", np.array([1]))
    m = np.zeros((1, 1, 100, 160), np.uint8); m[0, 0, 50:80, 30:60] = 1
    bl.add_field("mask", m)
    out = demo.overlay(image.copy(), bl)
    color = COCODemo.compute_colors_for_labels([1])[0]
    assert (out[20, 10:81] == color).all() and (out[20:91, 80] == color).all()          # box outline
    assert (out[60, 40] == (0.5 * color).astype(np.uint8)).all()                         # tinted mask
    text = (out[21:36, 11:75] > 180).all(-1)
    assert text.sum() > 30, "the class label is written at the box's top-left corner"


def test_bench_gpus_n_without_devices_fails_loudly():
    """`python bench.py --gpus 2` outside a torch.distributed environment starts its own two ranks; with no HIP device they fail,
    and the parent must exit non-zero without printing a JSON line (never a silent 1-GPU run reported as n_gpus=1)."""
    pytest.importorskip("torch")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # and inside a torch.distributed environment a --gpus / WORLD_SIZE mismatch is refused before anything else happens
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
