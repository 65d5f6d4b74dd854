"""RCCL path with world_size 1 on the GPU box (the only size a 1-GPU box allows): device-side record packing
equals the host-side layout, and the all-gather round-trips it.  N>1 is covered by the gloo CPU test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_world1_gather_matches_host_pack(ffi):
    from isegmi.dist import RcclGather, pack_records, record_bytes, unpack_records
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, fast_base_transform
    size, n = 200, 2
    net = Yolact(yolact_state_dict(1234), max_batch=n, input_size=size)
    rng = np.random.default_rng(3)
    x = fast_base_transform(rng.uniform(0, 255, (n, size, size, 3)).astype(np.float32))
    net(x)
    g = RcclGather(0, 1, RcclGather.unique_id(), record_bytes(n))
    g.gather_from(net)
    got = g.fetch()
    assert got.shape == (1, record_bytes(n))
    host = pack_records(net.fetch("det.count", n), net.fetch("det.box", n), net.fetch("det.score", n), net.fetch("det.class", n),
                        net.fetch("det.coeff", n))
    assert np.array_equal(got[0], host)
    rec = unpack_records(got[0], n)
    assert rec["count"].sum() > 0 and np.array_equal(rec["score"], net.fetch("det.score", n))
    # with the prototypes attached
    ph = net.fetch("proto", n).shape[1:3]
    g2 = RcclGather(0, 1, RcclGather.unique_id(), record_bytes(n, proto_hw=ph))
    g2.gather_from(net, with_proto=True)
    rec2 = unpack_records(g2.fetch()[0], n, proto_hw=ph)
    assert np.array_equal(rec2["proto"], net.fetch("proto", n))
    g.close(); g2.close(); net.close()


def test_rccl_world1_maskrcnn_records(ffi):
    from isegmi.dist import RcclGather, maskrcnn_record_bytes, pack_maskrcnn_records, unpack_maskrcnn_records
    from isegmi.maskrcnn import MaskRCNN, prepare_images
    from isegmi.weights import maskrcnn_state_dict
    rng = np.random.default_rng(2)
    x, hw = prepare_images([rng.uniform(0, 255, (200, 230, 3)).astype(np.float32)])
    model = MaskRCNN(maskrcnn_state_dict(1234), x.shape[1], x.shape[2], max_batch=1)
    model(x, hw)
    g = RcclGather(0, 1, RcclGather.unique_id(), maskrcnn_record_bytes(1))
    g.gather_from(model)
    got = g.fetch()[0]
    host = pack_maskrcnn_records(model.fetch("det.count", 1), model.fetch("det.box", 1), model.fetch("det.score", 1),
                                 model.fetch("det.label", 1), model.fetch("det.mask28", 1))
    assert np.array_equal(got, host)
    rec = unpack_maskrcnn_records(got, 1)
    assert rec["count"][0] > 0 and np.array_equal(rec["mask28"], model.fetch("det.mask28", 1))
    g.close(); model.close()


def test_rccl_world1_maskrcnn_c4_records(ffi):
    """The C4 predictor's 14x14 masks travel in the same record layout (M = 14)."""
    import dataclasses
    from isegmi.dist import RcclGather, maskrcnn_record_bytes, pack_maskrcnn_records, unpack_maskrcnn_records
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig, prepare_images
    from isegmi.weights import maskrcnn_c4_state_dict
    rng = np.random.default_rng(20261003)
    x, hw = prepare_images([rng.uniform(0, 255, (250, 340, 3)).astype(np.float32)], 16)
    cfg = dataclasses.replace(MaskRCNNConfig.c4(), RPN_POST_NMS_TOP_N_TEST=300)
    model = MaskRCNN(maskrcnn_c4_state_dict(1234), x.shape[1], x.shape[2], cfg=cfg, max_batch=1)
    model(x, hw)
    g = RcclGather(0, 1, RcclGather.unique_id(), maskrcnn_record_bytes(1, M=14))
    g.gather_from(model)
    got = g.fetch()[0]
    host = pack_maskrcnn_records(model.fetch("det.count", 1), model.fetch("det.box", 1), model.fetch("det.score", 1),
                                 model.fetch("det.label", 1), model.fetch("det.mask14", 1))
    assert np.array_equal(got, host)
    rec = unpack_maskrcnn_records(got, 1, M=14)
    assert rec["count"][0] > 0 and np.array_equal(rec["mask28"], model.fetch("det.mask14", 1))
    g.close(); model.close()


def test_rccl_queued_steps_without_host_sync(ffi):
    """Four forward -> gather_from steps on changing inputs, queued with no host synchronisation in between: the two record
    slots + the producer fence must keep every step's records intact (the last two are still fetchable: one per slot)."""
    from isegmi.dist import RcclGather, record_bytes, unpack_records
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, fast_base_transform
    size, n = 200, 2
    net = Yolact(yolact_state_dict(1234), max_batch=n, input_size=size)
    rng = np.random.default_rng(9)
    xs = [fast_base_transform(rng.uniform(0, 255, (n, size, size, 3)).astype(np.float32)) for _ in range(4)]
    clean = []
    for x in xs:  # clean, fully synchronised runs
        net(x)
        clean.append({k: net.fetch(k, n) for k in ("det.count", "det.score", "det.box", "det.class", "det.coeff")})
    assert not np.array_equal(clean[2]["det.score"], clean[3]["det.score"])
    ds = [ffi.DeviceBuffer.from_numpy(x) for x in xs]
    g = RcclGather(0, 1, RcclGather.unique_id(), record_bytes(n))
    for rep in range(3):  # 12 queued steps; the last pass ends on inputs 2, 3
        for d in ds:
            ffi.check(ffi.lib().isegmi_yolact_forward(net._h, d.ptr, n))
            g.gather_from(net)
    last = unpack_records(g.fetch()[0], n)
    prev = unpack_records(g.fetch(previous=True)[0], n)
    for rec, ref in ((last, clean[3]), (prev, clean[2])):
        assert np.array_equal(rec["count"], ref["det.count"])
        for i in range(n):
            c = int(ref["det.count"][i])
            assert np.array_equal(rec["score"][i, :c], ref["det.score"][i, :c])
            assert np.array_equal(rec["box"][i, :c], ref["det.box"][i, :c])
            assert np.array_equal(rec["cls"][i, :c], ref["det.class"][i, :c])
            assert np.array_equal(rec["coeff"][i, :c], ref["det.coeff"][i, :c])
    g.close(); net.close()


_TWO_RANK_WORKER = r'''
import json, os, sys
root, out = sys.argv[1], sys.argv[2]
sys.path[:0] = [root, os.path.join(root, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi.predictor import COCODemo, inference
from isegmi.weights import maskrcnn_state_dict
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
rng = np.random.default_rng(11)
images = [rng.integers(0, 256, s + (3,)).astype(np.uint8) for s in [(150, 200), (200, 150), (150, 200), (120, 200), (150, 200)]]
demo = COCODemo(None, min_image_size=160, confidence_threshold=0.0, state_dict=maskrcnn_state_dict(1234), max_image_size=288, max_batch=2, device=0)
try:
    res = inference(demo, images, batch_size=2, rank=rank, world=world)
except Exception as e:  # RCCL may refuse two ranks on one device
    open(out + ".err%d" % rank, "w").write(repr(e))
    raise
json.dump(res, open(out + ".%d" % rank, "w"))
'''


def test_two_ranks_share_one_gpu_through_rccl(ffi, tmp_path):
    """SURVEY 8(e) on the hardware at hand: TWO processes (ranks 0 and 1 of a world of 2, started by `python -m isegmi.launch`, TCP rendezvous, no
    torch) run the sharded inference() on the ONE visible GPU and all-gather their record blocks through RCCL; both must end with the complete
    result list, equal to the single-rank run.  An RCCL build that REFUSES two ranks on one device (its own error text) makes this a skip;
    anything else -- a wrong result, a crash, ranks that never get their communicator (the launcher's watchdog: exit 125) or never finish
    (exit 124) -- is a failure with both ranks' output attached (a time-out used to be recorded as a skip)."""
    import json
    import subprocess
    import sys
    import os
    from isegmi.predictor import COCODemo, inference
    from isegmi.weights import maskrcnn_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_TWO_RANK_WORKER)
    out = str(tmp_path / "res")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN", PYTHONPATH=os.path.join(root, "instancesegmentation-jittor_amd"))
    r = subprocess.run([sys.executable, "-m", "isegmi.launch", "--nproc", "2", "--timeout", "240", "--init-timeout", "120", str(script), root, out],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    log = r.stdout.decode(errors="replace")
    if r.returncode != 0:
        refused = ("uplicate GPU" in log) or ("invalid usage" in log.lower()) or any(
            os.path.exists(out + ".err%d" % k) and ("invalid usage" in open(out + ".err%d" % k).read().lower() or "ncclCommInitRank" in open(out + ".err%d" % k).read())
            for k in range(2))
        if refused and r.returncode not in (124, 125):
            pytest.skip("this RCCL refuses two ranks on one device: " + log[-300:])
        pytest.fail("two-rank run failed (rc %d; 124 = time limit, 125 = no communicator within the watchdog's limit):\n%s" % (r.returncode, log[-3000:]))
    got = [json.load(open(out + ".%d" % k)) for k in range(2)]
    rng = np.random.default_rng(11)
    images = [rng.integers(0, 256, s + (3,)).astype(np.uint8) for s in [(150, 200), (200, 150), (150, 200), (120, 200), (150, 200)]]
    demo = COCODemo(None, min_image_size=160, confidence_threshold=0.0, state_dict=maskrcnn_state_dict(1234), max_image_size=288, max_batch=2)
    want = inference(demo, images, batch_size=2)
    demo.close()
    assert got[0] == got[1] == json.loads(json.dumps(want)) and len(want) > 20


@pytest.mark.parametrize("launcher", ["isegmi.launch", "env"])
@pytest.mark.parametrize("model_args", [["--no-maskrcnn"], ["--model", "maskrcnn"]])
def test_bench_multi_rank_code_path_on_one_gpu(ffi, model_args, launcher):
    """bench.py's N > 1 code path on the one GPU at hand (one rank, ISEGMI_BENCH_FORCE_DIST=1): started by the package's launcher (what
    `bench.py --gpus N` from a plain shell does) and from a bare rank environment (RANK / WORLD_SIZE / MASTER_*: what the driver's
    `python -m torch.distributed.run` hands every rank) -- TCP rendezvous, three RCCL communicators (the 8-byte control gather behind barrier
    and MAX, the raw-record gather inside `value`, the COCO-record gather inside `value_e2e`), ONE JSON line; bench.py imports no torch."""
    import json
    import os
    import re
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert not re.search(r"^\s*(import|from)\s+torch", open(os.path.join(root, "bench.py")).read(), re.M)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    args = [os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-latency"] + model_args
    env = dict(os.environ, ISEGMI_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=os.path.join(root, "instancesegmentation-jittor_amd"))
    if launcher == "isegmi.launch":
        cmd = [sys.executable, "-m", "isegmi.launch", "--nproc", "1", "--timeout", "500", "--init-timeout", "200"] + args
    else:
        cmd = [sys.executable] + args
        env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID="t")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["ranks_with_records"] == 1 and d["value"] > 0 and d["value_e2e"] > 0
    assert d["e2e"]["device_rle_equals_host_encoder"] is True and d["e2e"]["blocks_collected"] == 6
