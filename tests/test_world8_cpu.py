"""The N = 8 control path, rehearsed on the CPU (no 8-GPU node has ever run this code, so everything that does not need
a GPU is exercised here at the world size the driver will use): the TCP rendezvous with eight ranks, two jobs whose port windows overlap,
`python -m isegmi.launch --nproc 8`, and `inference()`'s global schedule -- round-robin batches, the empty step of a rank without a batch,
the RLE-overflow redo that every rank has to take in step -- with eight ranks over a fake all-gather.  The record blocks are the real ones
(isegmi.dist.pack_coco_records / unpack_coco_records), the engine and the transport are stand-ins."""
import multiprocessing as mp
import os
import random
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "instancesegmentation-jittor_amd")


def _free_port_base():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _rank(rank, world, port, run_id, first_byte, calls, delay, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ISEGMI_RUN_ID=run_id)
    os.environ.pop("TORCHELASTIC_RUN_ID", None)
    time.sleep(delay)
    from isegmi.dist import rendezvous_unique_id
    for c in range(calls):
        uid = rendezvous_unique_id(rank, world, lambda c=c: bytes([first_byte, c]) + bytes(range(2, 128)), timeout=60.0)
        q.put((port, rank, c, uid))


def test_rendezvous_world8_shuffled_start_two_calls():
    """eight ranks started in a shuffled order with staggered delays (rank 0 neither first nor last), two rendezvous in a row (a job makes one per
    communicator; bench.py makes exactly one since round 5): every rank gets rank 0's id of the matching call"""
    port = _free_port_base()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    order = list(range(8))
    random.Random(5).shuffle(order)
    order.remove(0); order.insert(4, 0)
    ps = [ctx.Process(target=_rank, args=(r, 8, port, "w8", 0xA0, 2, 0.05 * i, q)) for i, r in enumerate(order)]
    for p in ps:
        p.start()
    got = [q.get(timeout=120) for _ in range(16)]
    for p in ps:
        p.join(60)
    assert all(p.exitcode == 0 for p in ps)
    assert sorted((r, c) for _, r, c, _ in got) == [(r, c) for r in range(8) for c in range(2)]
    assert all(uid == bytes([0xA0, c]) + bytes(range(2, 128)) for _, _, c, uid in got)


def test_two_jobs_with_one_run_id_on_adjacent_ports_keep_their_own_ids():
    """under `python -m torch.distributed.run` every job's run id is "none"; two jobs of equal world size on MASTER_PORT p and p + 1
    have overlapping rendezvous windows.  Job B's rank 0 is listening first (on p + 2, the first port of its window and the SECOND of job A's);
    job A's ranks 1, 2 start before A's rank 0, walk p + 1 (nobody yet) and p + 2 (job B) -- and must not take B's id: the hello carries the
    job's MASTER_PORT and the answer must echo it."""
    p = _free_port_base()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    b0 = ctx.Process(target=_rank, args=(0, 3, p + 1, "none", 0xBB, 1, 0.0, q))
    b0.start()
    time.sleep(1.0)                      # B's rank 0 has bound p + 2
    a12 = [ctx.Process(target=_rank, args=(r, 3, p, "none", 0xAA, 1, 0.0, q)) for r in (1, 2)]
    for x in a12:
        x.start()
    time.sleep(1.0)                      # A's ranks have been knocking on B's door for a second
    rest = [ctx.Process(target=_rank, args=(0, 3, p, "none", 0xAA, 1, 0.0, q))] + [ctx.Process(target=_rank, args=(r, 3, p + 1, "none", 0xBB, 1, 0.0, q)) for r in (1, 2)]
    for x in rest:
        x.start()
    got = [q.get(timeout=120) for _ in range(6)]
    for x in [b0] + a12 + rest:
        x.join(60)
    for port, rank, _, uid in got:
        assert uid[0] == (0xAA if port == p else 0xBB), "rank %d of the job on port %d got the other job's id" % (rank, port)


def test_launcher_nproc8_environment_and_exit_code(tmp_path):
    """python -m isegmi.launch --nproc 8: eight fresh processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the rendezvous works between
    them, a failing rank's code is the launcher's code and takes the other seven down"""
    w = tmp_path / "w.py"
    w.write_text("import os, sys, time\nsys.path.insert(0, %r)\nfrom isegmi.dist import rendezvous_unique_id, notify_launcher\n"
                 "r, n = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\nassert int(os.environ['LOCAL_RANK']) == r and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                 "uid = rendezvous_unique_id(r, n, lambda: bytes([9]) * 128, timeout=60)\nnotify_launcher('comm %%d' %% r)\n"
                 "open(os.path.join(%r, 'ok%%d' %% r), 'wb').write(uid + bytes([n]))\n"
                 "if len(sys.argv) > 1 and sys.argv[1] == 'fail':\n"
                 "    if r == 5:\n        sys.exit(7)\n    time.sleep(60)\n" % (PKG, str(tmp_path)))
    env = dict(os.environ, PYTHONPATH=PKG)
    r = subprocess.run([sys.executable, "-m", "isegmi.launch", "--nproc", "8", "--init-timeout", "90", str(w)], env=env, timeout=180)
    assert r.returncode == 0
    assert all((tmp_path / ("ok%d" % i)).read_bytes() == bytes([9]) * 128 + bytes([8]) for i in range(8))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "isegmi.launch", "--nproc", "8", str(w), "fail"], env=env, timeout=180)
    assert r.returncode == 7 and time.time() - t0 < 50, "rank 5's exit code, and the sleeping ranks were killed with it"


# ------------------------------------------------------------------------------------------------ inference() at world 8, fake engine / transport
K = 100


class _Exchange:
    """eight threads' all-gather: everybody deposits a block, everybody reads all of them"""

    def __init__(self, world):
        self.world = world
        self.blocks = [None] * world
        self.bar = threading.Barrier(world, timeout=60)
        self.sizes = []

    def allgather(self, rank, block):
        self.blocks[rank] = block
        self.bar.wait()
        assert len({b.size for b in self.blocks}) == 1, "ranks disagree about the block size of a step"
        out = np.concatenate(self.blocks)
        if rank == 0:
            self.sizes.append(block.size)
        self.bar.wait()
        return out


class _FakeGather:
    def __init__(self, ex, rank, nbytes):
        self.ex, self.rank, self.world, self.nbytes = ex, rank, ex.world, int(nbytes)
        self.recv = None
        self.empty_steps = self.resizes = 0
        self.log = []   # RcclGather.log: (kind, bytes, stream) of every collective this rank issued

    def gather_coco_from(self, net, n_block):
        b = net.pack(n_block)
        assert b.size == self.nbytes
        self.log.append(("data", b.size, "results"))
        self.recv = self.ex.allgather(self.rank, b)

    def gather_empty(self, net=None):
        self.empty_steps += 1
        self.log.append(("empty", self.nbytes, "results" if net is not None else "own"))
        self.recv = self.ex.allgather(self.rank, np.zeros(self.nbytes, np.uint8))

    def fence_results_stream(self, net):
        pass

    def resize(self, nbytes):
        self.resizes += 1
        self.nbytes = int(nbytes)

    def wait(self):
        pass

    def close(self):
        pass


class _Cfg:
    det_cap = K
    SIZE_DIVISIBILITY = 32


class _FakeModel:
    """What inference() and RecordPipeline touch of a MaskRCNN engine.  An image filled with the value v "detects" v % 4 objects whose RLE strings
    are 3 v + k + 1 characters long, so string volume grows with the image index and the small initial capacity overflows on the way."""
    KIND = 2
    H = W = 64

    def __init__(self, bs):
        self.cfg, self.max_batch = _Cfg(), bs
        self.cap_chars = 40 * bs          # per block
        self.vals, self.sparse = [], None
        self.redone = 0

    def _u8_staging(self, slot, n):
        pass

    def upload_u8_async(self, pin, hw, slot):
        off, self.vals = 0, []
        for h, w in hw:
            self.vals.append(int(pin.array[off]))
            off += h * w * 3

    def forward_device(self, n, slot=0):
        assert n == len(self.vals)

    def paste_device(self, h, w, sizes=None):
        pass

    def rle_device(self, oh=None):
        pass

    def set_param(self, name, v):
        if name == "rle_cap_chars":
            self.cap_chars = int(v)
            self.redone += 1
        elif name == "sparse_masks":
            self.sparse = v

    def coco_record_bytes(self, n):
        from isegmi.dist import coco_record_layout
        _, coff, total = coco_record_layout(n, K, 2, False, self.cap_chars)
        return total, coff

    @staticmethod
    def detections(v):
        c = v % 4
        box = np.array([[v, k, v + 10 + k, k + 20] for k in range(c)], np.float32).reshape(c, 4)
        return c, box, np.array([0.9 - 0.1 * k for k in range(c)], np.float32), np.array([1 + (v + k) % 80 for k in range(c)], np.int32), [b"a" * (3 * v + k + 1) for k in range(c)]

    def pack(self, n_block):
        from isegmi.dist import pack_coco_records
        cnt = np.zeros(n_block, np.int32); box = np.zeros((n_block, K, 4), np.float32); sc = np.zeros((n_block, K), np.float32)
        lab = np.zeros((n_block, K), np.int32); so = np.zeros(n_block * K + 1, np.int32)
        chars, pos = b"", 0
        for i in range(n_block):
            strs = []
            if i < len(self.vals):
                c, b, s, l, strs = self.detections(self.vals[i])
                cnt[i] = c; box[i, :c] = b; sc[i, :c] = s; lab[i, :c] = l
            for k in range(K):
                so[i * K + k] = pos
                if k < len(strs):
                    chars += strs[k]; pos += len(strs[k])
        so[n_block * K] = pos
        if pos > self.cap_chars:      # the device flags the overflow and ships no strings
            return pack_coco_records([17, pos, 2, 0], box, cnt, sc, lab, np.zeros_like(so), b"", n_block, K, 2, None, self.cap_chars)
        return pack_coco_records([17, pos, 0, 0], box, cnt, sc, lab, so, chars, n_block, K, 2, None, self.cap_chars)

    def download_async(self, slot, pin, src, nbytes):
        pin.array[:nbytes] = src[:nbytes]

    def download_wait(self, slot):
        pass

    def sync(self):
        pass


class _Pinned:
    def __init__(self, shape, dtype=np.float32):
        self.array = np.zeros(shape, dtype)
        self.nbytes = self.array.nbytes

    def free(self):
        pass


class _Predictor:
    min_image_size, max_image_size = 32, 64
    cfg = _Cfg()

    def __init__(self, bs):
        self.max_batch = bs
        self.model = _FakeModel(bs)

    def engine(self, bs):
        return self.model


@pytest.mark.parametrize("n_img,bs", [(37, 2), (5, 1), (16, 2)])
def test_inference_schedule_world8_with_empty_steps_and_overflow_redo(monkeypatch, n_img, bs):
    """eight ranks, an image list that does not divide over them (37 images in batches of 2 = 19 batches: ranks 3-7 sit out the last step; 5
    images: ranks 5-7 never have a batch), RLE capacities that overflow on the way (all ranks must grow and re-size in the same step, or the
    all-gather of the next step has blocks of different sizes: _Exchange asserts).  Every rank returns the complete, identical result list."""
    from isegmi import _ffi, pipeline, predictor
    world = 8
    ex = _Exchange(world)
    monkeypatch.setattr(_ffi, "PinnedBuffer", _Pinned)
    gathers = {}

    def make_gather(net, batch, rank, world_, force=False):
        gathers[rank] = _FakeGather(ex, rank, net.coco_record_bytes(batch)[0])
        return gathers[rank]
    monkeypatch.setattr(pipeline, "make_gather", make_gather)
    images = [np.full((32, 48, 3), i, np.uint8) for i in range(n_img)]
    results, errors, preds = [None] * world, [], [_Predictor(bs) for _ in range(world)]

    def run(rank):
        try:
            results[rank] = predictor.inference(preds[rank], images, image_ids=[100 + i for i in range(n_img)], batch_size=bs, rank=rank, world=world, workers=0)
        except BaseException as e:   # noqa: BLE001 -- reported below; a dead rank must not leave the others in the barrier for a minute
            errors.append((rank, repr(e)))
            ex.bar.abort()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errors, errors
    want = []
    for i in range(n_img):
        c, box, sc, lab, strs = _FakeModel.detections(i)
        from isegmi.coco import COCO_CATEGORY_IDS
        for k in range(c):
            b = box[k].astype(np.float64)
            want.append({"image_id": 100 + i, "category_id": COCO_CATEGORY_IDS[int(lab[k]) - 1], "bbox": [b[0], b[1], b[2] - b[0] + 1.0, b[3] - b[1] + 1.0],
                         "score": float(np.float64(sc[k])), "segmentation": {"size": [32, 48], "counts": strs[k].decode()}})
    for r in range(world):
        assert results[r] == want, "rank %d" % r
    nbatches = -(-n_img // bs)
    nsteps = -(-nbatches // world)
    for r in range(world):
        mine = len(range(r, nbatches, world))
        assert gathers[r].empty_steps >= nsteps - mine                # (a redone step of an idle rank is one more empty block)
        assert preds[r].model.sparse == 0.0                            # the pipeline was closed on the way out
        assert preds[r].model.redone == preds[0].model.redone and gathers[r].resizes == gathers[0].resizes
    # the sequence of collectives -- how many, of how many bytes each -- is the same on all eight ranks (a rank without a batch issues
    # an empty block where the others issue data), and every one of them, empty steps and redone steps included, goes to the results stream
    seq0 = [b for _, b, _ in gathers[0].log]
    assert len(seq0) >= nsteps
    for r in range(world):
        assert [b for _, b, _ in gathers[r].log] == seq0, "rank %d issued another sequence of collectives" % r
        assert {s for _, _, s in gathers[r].log} == {"results"}, "rank %d put a collective on a second stream" % r
        assert sum(k == "empty" for k, _, _ in gathers[r].log) == gathers[r].empty_steps
    if n_img == 37:
        assert any(k == "empty" for k, _, _ in gathers[7].log) and all(k == "data" for k, _, _ in gathers[0].log)
    if n_img >= 16:
        assert preds[0].model.redone >= 1, "the fixture is meant to overflow the initial capacity"
        assert len(set(ex.sizes)) >= 2, "the block size grew in step on every rank"


# ------------------------------------------------------------------------------------------------ bench.py's N > 1 harness at world 8 (no GPU)
_BENCH_WORKER = r'''
import os, sys, types
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np
import torch, torch.distributed as td
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
td.init_process_group("gloo", init_method="tcp://127.0.0.1:%(gloo_port)d", rank=rank, world_size=world)
from isegmi import _ffi, dist as idist
_ffi.set_device = lambda i: None          # no HIP device here: the harness logic is what runs
log = []
class FakeGather:                          # RcclGather's surface as bench.Dist uses it, on gloo
    SLOTS = 2
    made = 0
    def __init__(self, rank, world, uid, nbytes):
        assert len(uid) == 128
        FakeGather.made += 1
        self.rank, self.world, self.nbytes, self.capacity, self.init_seconds, self.uid = rank, world, int(nbytes), int(nbytes), 0.01, bytes(uid)
    @staticmethod
    def unique_id():
        return bytes([42]) * 128
    def allgather_bytes(self, data):
        a = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        assert 0 < a.numel() <= self.capacity
        out = [torch.empty_like(a) for _ in range(self.world)]
        td.all_gather(out, a)
        log.append(("ctl", a.numel()))
        return np.stack([o.numpy() for o in out])
    def resize(self, n):
        self.nbytes = int(n); self.capacity = max(self.capacity, self.nbytes); log.append(("resize", int(n)))
    def wait(self): pass
    def close(self): log.append(("close",))
idist.RcclGather = FakeGather
import bench
a = types.SimpleNamespace(gpus=world)
d = bench.Dist(a)
assert d.on and d.rank == rank and d.world == world
d.barrier()
m = d.max(1.0 + rank)                      # MAX over ranks of the wall time
assert m == float(world), m
g1 = d.make_gather(123456)                 # the record blocks of the timed loops
g2 = d.make_gather(7890123)                # the COCO blocks of the end-to-end loop: the SAME communicator, re-sized
assert g1 is g2 and g1.nbytes == 7890123 and FakeGather.made == 1, "one communicator per rank"
assert g1.uid == bytes([42]) * 128         # rank 0's id reached this rank through the TCP rendezvous
info = d.info()
assert info["rccl_communicators_per_rank"] == 1 and [r["rank"] for r in info["ranks"]] == list(range(world))
assert [r["device_id"] for r in info["ranks"]] == list(range(world))
d.barrier()
d.close()
assert log[-1] == ("close",)
# every rank issued the same sequence of collectives and re-sizes
seq = repr([x for x in log]).encode()
allseq = [None] * world
td.all_gather_object(allseq, seq)
assert len(set(allseq)) == 1, "ranks disagree about the order of collectives"
open(os.path.join(%(out)r, "ok%%d" %% rank), "w").write("ok")
td.destroy_process_group()
'''


def test_bench_dist_harness_world8_one_communicator(tmp_path):
    """bench.py's N > 1 harness (class Dist) at the world size the driver will use, on gloo: the ranks meet through the package's own TCP rendezvous ONCE,
    hold ONE communicator each (control words, record blocks and COCO blocks share it), issue the same sequence of collectives, and the N > 1 JSON keys
    (`rccl_communicators_per_rank`, `ranks`) come out as documented.  (the three-communicator harness had never run with more than one rank.)"""
    pytest.importorskip("torch")
    w = tmp_path / "w.py"
    w.write_text(_BENCH_WORKER % dict(root=ROOT, pkg=PKG, gloo_port=_free_port_base(), out=str(tmp_path)))
    env = dict(os.environ, PYTHONPATH=PKG, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "isegmi.launch", "--nproc", "8", "--timeout", "240", str(w)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert all((tmp_path / ("ok%d" % i)).exists() for i in range(8))
