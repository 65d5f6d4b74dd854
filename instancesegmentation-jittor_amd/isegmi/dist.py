"""Batch-sharded multi-GPU inference: one process per GPU, one all-gather of detection records.

SURVEY.md 8(e): images are independent, so a global batch is split contiguously over ranks and the only
exchange is an all-gather of fixed-capacity records (upstream analogue: the pickle all_gather of
{image_id: BoxList} in maskrcnn-benchmark's engine/inference.py, reached from README.md:344-347).
The record layout is defined once here (pack_records/unpack_records) and produced on the device by
isegmi_yolact_pack_records; the transport is RCCL (GPU) or any callable (gloo in the CPU tests).
"""
import ctypes as C

import numpy as np

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


def gather_records(local_record, world, allgather):
    """allgather(np.uint8[nbytes]) -> list of `world` arrays.  Returns them in rank order."""
    parts = allgather(local_record)
    assert len(parts) == world
    return parts


class RcclGather:
    """RCCL all-gather of the engine's packed detection records (device side).

    Two (send, recv) slots alternate between steps: step t packs into / gathers through slot t % 2 while step t-1's
    all-gather may still be in flight on the comm stream.  Before a slot is re-packed the producer stream waits (on the
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
        """Pack the last forward's records on the engine's results stream, then all-gather on the comm stream (async)."""
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
