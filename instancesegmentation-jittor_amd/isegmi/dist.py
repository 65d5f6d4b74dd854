"""Batch-sharded multi-GPU inference: one process per GPU, one all-gather of detection records.

SURVEY.md 8(e): images are independent, so a global batch is split contiguously over ranks and the only
exchange is an all-gather of fixed-capacity records (upstream analogue: the pickle all_gather of
{image_id: BoxList} in maskrcnn-benchmark's engine/inference.py, reached from README.md:344-347).
The record layout is defined once here (pack_records/unpack_records) and produced on the device by
isegmi_yolact_pack_records; the transport is RCCL (GPU) or any callable (gloo in the CPU tests).
"""
import ctypes as C
import time as _time

import numpy as np

_T0 = _time.time()  # about when this process started: a rendezvous file much older than that is a crashed run's left-over

from . import _ffi

K_DEFAULT = 100
MD = 32


def shard_batch(global_batch, rank, world):
    """Contiguous split of `global_batch` images: returns (start, stop) for `rank`."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def record_bytes(n, K=K_DEFAULT, md=MD, proto_hw=None):
    b = n * 4 + n * K * 16 + n * K * 4 + n * K * 4 + n * K * md * 4
    if proto_hw:
        b += n * proto_hw[0] * proto_hw[1] * md * 4
    return b


def pack_records(count, box, score, cls, coeff, proto=None):
    """Host-side restatement of the device record layout (used by tests and by unpack round trips)."""
    parts = [np.ascontiguousarray(count, np.int32), np.ascontiguousarray(box, np.float32),
             np.ascontiguousarray(score, np.float32), np.ascontiguousarray(cls, np.int32),
             np.ascontiguousarray(coeff, np.float32)]
    if proto is not None:
        parts.append(np.ascontiguousarray(proto, np.float32))
    return np.concatenate([p.view(np.uint8).ravel() for p in parts])


def unpack_records(buf, n, K=K_DEFAULT, md=MD, proto_hw=None):
    buf = np.ascontiguousarray(buf, np.uint8)
    off = 0

    def take(nbytes, dtype, shape):
        nonlocal off
        a = buf[off:off + nbytes].view(dtype).reshape(shape)
        off += nbytes
        return a
    out = dict(count=take(n * 4, np.int32, (n,)), box=take(n * K * 16, np.float32, (n, K, 4)),
               score=take(n * K * 4, np.float32, (n, K)), cls=take(n * K * 4, np.int32, (n, K)),
               coeff=take(n * K * md * 4, np.float32, (n, K, md)))
    if proto_hw:
        out["proto"] = take(n * proto_hw[0] * proto_hw[1] * md * 4, np.float32, (n, proto_hw[0], proto_hw[1], md))
    return out


def maskrcnn_record_bytes(n, K=K_DEFAULT, M=28):
    """M = RoI mask resolution: 28 (FPN mask head), 14 (MaskRCNNC4Predictor)."""
    return n * 4 + n * K * 16 + n * K * 4 + n * K * 4 + n * K * M * M * 4


def pack_maskrcnn_records(count, box, score, label, mask28):
    parts = [np.ascontiguousarray(count, np.int32), np.ascontiguousarray(box, np.float32), np.ascontiguousarray(score, np.float32),
             np.ascontiguousarray(label, np.int32), np.ascontiguousarray(mask28, np.float32)]
    return np.concatenate([p.view(np.uint8).ravel() for p in parts])


def unpack_maskrcnn_records(buf, n, K=K_DEFAULT, M=28):
    buf = np.ascontiguousarray(buf, np.uint8)
    sizes = [(n * 4, np.int32, (n,)), (n * K * 16, np.float32, (n, K, 4)), (n * K * 4, np.float32, (n, K)), (n * K * 4, np.int32, (n, K)),
             (n * K * M * M * 4, np.float32, (n, K, M, M))]
    out, off = [], 0
    for nb, dt, shp in sizes:
        out.append(buf[off:off + nb].view(dt).reshape(shp)); off += nb
    return dict(zip(("count", "box", "score", "label", "mask28"), out))


# ------------------------------------------------------------------------------------------------ COCO record block (device-side output)
def coco_record_layout(n, K=K_DEFAULT, kind=2, has_mscore=False, cap_chars=0):
    """Byte layout of the fixed-size block isegmi_engine_pack_coco_records writes for a batch of n images (csrc/results.cpp):
    -> (sections {name: (offset, nbytes, dtype, shape)}, chars_offset, total_bytes).  kind: 1 Yolact (int64 boxes), 2 Mask R-CNN (fp32)."""
    y = kind == 1
    secs, off = {}, 0
    for name, nb, dt, shp in (("status", 16, np.int32, (4,)),
                              ("box", n * K * (32 if y else 16), np.int64 if y else np.float32, (n, K, 4)),
                              ("count", n * 4, np.int32, (n,)),
                              ("score", n * K * 4, np.float32, (n, K)),
                              ("label", n * K * 4, np.int32, (n, K))) + \
                             ((("mscore", n * K * 4, np.float32, (n, K)),) if (y and has_mscore) else ()) + \
                             (("str_off", (n * K + 1) * 4, np.int32, (n * K + 1,)),):
        secs[name] = (off, nb, dt, shp)
        off += nb
    chars_off = (off + 7) & ~7
    return secs, chars_off, chars_off + int(cap_chars)


class CocoRecordError(RuntimeError):
    pass


def unpack_coco_records(buf, n, K=K_DEFAULT, kind=2, has_mscore=False, cap_chars=0):
    """One rank's block -> dict(status, box, count, score, label[, mscore], str_off, chars=bytes).  Raises if the engine flagged an
    overflow of its RLE capacities (the strings would be incomplete): raise the engine's rle_cap_runs / rle_cap_chars."""
    buf = np.ascontiguousarray(buf, np.uint8).ravel()
    secs, coff, total = coco_record_layout(n, K, kind, has_mscore, cap_chars)
    assert buf.size >= total, (buf.size, total)
    out = {name: buf[off:off + nb].view(dt).reshape(shp) for name, (off, nb, dt, shp) in secs.items()}
    st = out["status"]
    if st[2] != 0:
        raise CocoRecordError("device RLE overflow (%s): %d runs, %d characters; raise the engine's rle_cap_runs / rle_cap_chars" %
                              ("runs" if st[2] & 1 else "characters", int(st[0]), int(st[1])))
    nchars = int(out["str_off"][-1])
    assert nchars == int(st[1]) and nchars <= cap_chars, (nchars, int(st[1]), cap_chars)
    out["chars"] = buf[coff:coff + nchars].tobytes()
    return out


def pack_coco_records(status, box, count, score, label, str_off, chars, n, K=K_DEFAULT, kind=2, mscore=None, cap_chars=0):
    """Host-side restatement of the block (tests; round trips through unpack_coco_records)."""
    secs, coff, total = coco_record_layout(n, K, kind, mscore is not None, cap_chars)
    buf = np.zeros(total, np.uint8)
    vals = dict(status=status, box=box, count=count, score=score, label=label, str_off=str_off)
    if mscore is not None:
        vals["mscore"] = mscore
    for name, (off, nb, dt, shp) in secs.items():
        buf[off:off + nb] = np.ascontiguousarray(vals[name], dt).reshape(shp).view(np.uint8).ravel()
    c = np.frombuffer(bytes(chars), np.uint8)
    buf[coff:coff + c.size] = c
    return buf


def gather_records(local_record, world, allgather):
    """allgather(np.uint8[nbytes]) -> list of `world` arrays.  Returns them in rank order."""
    parts = allgather(local_record)
    assert len(parts) == world
    return parts


class RcclGather:
    """RCCL all-gather of the engine's packed detection records (device side).

    Two (send, recv) slots alternate between steps: step t packs into / gathers through slot t % 2 while step t-1's
    all-gather may still be in flight on the engine's results stream.  Before a slot is re-packed the producer stream waits (on the
    device, isegmi_comm_fence_producer) for that slot's previous all-gather, so consecutive `gather_from` calls need no
    host synchronisation in between and ranks never see records of mixed steps.  `fetch()` returns the records of the
    most recent `gather_from`; `fetch(previous=True)` those of the one before (still intact: the other slot)."""

    SLOTS = 2

    def __init__(self, rank, world, uid_bytes, per_rank_bytes):
        self.rank, self.world, self.nbytes = rank, world, int(per_rank_bytes)
        self._c = C.c_void_p()
        uid = C.create_string_buffer(bytes(uid_bytes), 128)
        _ffi.check(_ffi.lib().isegmi_comm_create(uid, rank, world, C.byref(self._c)))
        self.sends = [_ffi.DeviceBuffer((self.nbytes,), np.uint8) for _ in range(self.SLOTS)]
        self.recvs = [_ffi.DeviceBuffer((self.nbytes * world,), np.uint8) for _ in range(self.SLOTS)]
        self.step = 0

    @staticmethod
    def unique_id():
        b = C.create_string_buffer(128)
        _ffi.check(_ffi.lib().isegmi_comm_unique_id(b))
        return b.raw

    @property
    def send(self):  # slot of the most recent gather
        return self.sends[(self.step - 1) % self.SLOTS]

    @property
    def recv(self):
        return self.recvs[(self.step - 1) % self.SLOTS]

    def gather_from(self, net, with_proto=False):
        """Pack the last forward's records on the engine's results stream, then all-gather right behind it on the same stream (async)."""
        L = _ffi.lib()
        slot = self.step % self.SLOTS
        send, recv = self.sends[slot], self.recvs[slot]
        st = C.c_void_p()
        _ffi.check(L.isegmi_engine_stream(net._h, C.byref(st)))
        _ffi.check(L.isegmi_comm_fence_producer(self._c, slot, st))  # WAR: the slot's previous gather has finished
        nb = C.c_int64()
        if getattr(net, "KIND", 1) == 2:
            _ffi.check(L.isegmi_maskrcnn_pack_records(net._h, send.ptr, C.c_int64(self.nbytes), C.byref(nb)))
        else:
            _ffi.check(L.isegmi_yolact_pack_records(net._h, send.ptr, C.c_int64(self.nbytes), int(with_proto), C.byref(nb)))
        assert nb.value == self.nbytes, (nb.value, self.nbytes)
        _ffi.check(L.isegmi_engine_stream(net._h, C.byref(st)))
        _ffi.check(L.isegmi_comm_allgather_slot(self._c, slot, send.ptr, recv.ptr, C.c_int64(self.nbytes), st))
        self.step += 1

    def gather_coco_from(self, net, n_block):
        """The device-side COCO output of the last step (isegmi_engine_pack_coco_records: boxes in original-image coordinates, scores,
        labels, RLE strings) through the same two-slot all-gather; per_rank_bytes must be net.coco_record_bytes(n_block)."""
        L = _ffi.lib()
        slot = self.step % self.SLOTS
        send, recv = self.sends[slot], self.recvs[slot]
        st = C.c_void_p()
        _ffi.check(L.isegmi_engine_stream(net._h, C.byref(st)))
        _ffi.check(L.isegmi_comm_fence_producer(self._c, slot, st))
        nb = C.c_int64()
        _ffi.check(L.isegmi_engine_pack_coco_records(net._h, send.ptr, C.c_int64(self.nbytes), int(n_block), C.byref(nb)))
        assert nb.value == self.nbytes, (nb.value, self.nbytes)
        _ffi.check(L.isegmi_engine_stream(net._h, C.byref(st)))
        _ffi.check(L.isegmi_comm_allgather_slot(self._c, slot, send.ptr, recv.ptr, C.c_int64(self.nbytes), st))
        self.step += 1

    def gather_empty(self):
        """A step in which this rank has no batch (the image list does not divide over the ranks): an all-zero block -- every count 0."""
        L = _ffi.lib()
        slot = self.step % self.SLOTS
        _ffi.check(L.isegmi_comm_wait_slot(self._c, slot))
        self.sends[slot].zero()  # synchronous memset: rare (at most once per rank and data set)
        _ffi.check(L.isegmi_comm_allgather_slot(self._c, slot, self.sends[slot].ptr, self.recvs[slot].ptr, C.c_int64(self.nbytes), None))
        self.step += 1

    def fence_results_stream(self, net):
        """Later work on the engine's results stream waits (on the device) for the most recent gather's slot."""
        st = C.c_void_p()
        _ffi.check(_ffi.lib().isegmi_engine_stream(net._h, C.byref(st)))
        _ffi.check(_ffi.lib().isegmi_comm_fence_producer(self._c, (self.step - 1) % self.SLOTS, st))

    def wait(self):
        _ffi.check(_ffi.lib().isegmi_comm_wait(self._c))

    def fetch(self, previous=False):
        assert self.step > (1 if previous else 0), "fetch before gather_from"
        slot = (self.step - (2 if previous else 1)) % self.SLOTS
        _ffi.check(_ffi.lib().isegmi_comm_wait_slot(self._c, slot))
        return self.recvs[slot].numpy().reshape(self.world, self.nbytes)

    def close(self):
        if self._c:
            _ffi.lib().isegmi_comm_destroy(self._c)
            self._c = None


# ------------------------------------------------------------------------------------------------ rendezvous without torch
def rendezvous_unique_id(rank, world, make_uid, timeout=120.0):
    """Ships rank 0's 128-byte RCCL unique id to every rank of ONE node through a file (SURVEY 8e / north_star: the ranks are the GPUs of
    one node).  The file name comes from the launcher's environment (MASTER_PORT + TORCHELASTIC_RUN_ID, as `python -m
    torch.distributed.run` and bench.py's own spawner set them), so concurrent jobs do not collide; ISEGMI_RDZV_DIR overrides the
    directory (default: /dev/shm, else the system temp dir).  Stdlib only: the package does not import torch."""
    import os
    import tempfile
    import time
    if world == 1:
        return make_uid()
    d = os.environ.get("ISEGMI_RDZV_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    tag = "%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", os.environ.get("ISEGMI_RUN_ID", "0")))
    path = os.path.join(d, "isegmi_uid_%s_%d" % (tag.replace("/", "_"), world))
    if rank == 0:
        uid = make_uid()
        tmp = path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)  # atomic: a reader sees all 128 bytes or no file
        return uid
    t0 = time.time()
    while True:
        try:
            if os.path.getmtime(path) >= _T0 - 60.0:  # the launcher starts all ranks together; older = stale (rank 0 replaces it)
                with open(path, "rb") as f:
                    uid = f.read()
                if len(uid) == 128:
                    return uid
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError("rank %d: no RCCL unique id at %s after %.0f s (is rank 0 alive?)" % (rank, path, timeout))
        time.sleep(0.01)


def rendezvous_cleanup(rank, world):
    import os
    import tempfile
    if world == 1 or rank != 0:
        return
    d = os.environ.get("ISEGMI_RDZV_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir())
    tag = "%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", os.environ.get("ISEGMI_RUN_ID", "0")))
    try:
        os.remove(os.path.join(d, "isegmi_uid_%s_%d" % (tag.replace("/", "_"), world)))
    except OSError:
        pass
