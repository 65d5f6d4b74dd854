"""Step pipeline of the device-side COCO output: every step's detections leave the GPU as ONE fixed-size record block (boxes in
original-image coordinates, scores, labels, pycocotools RLE strings; csrc/results.cpp) -- downloaded asynchronously into pinned memory
(one rank) or all-gathered over RCCL (several ranks) -- while the next step already runs.  `submit()` hands back the records
of the PREVIOUS step, so the host unpacks step t-1 while the GPU computes step t.  Used by isegmi.predictor.inference (Mask R-CNN,
tools/test_net.py path, README.md:344-347), isegmi.yolact.evaluate (eval.py path, README.md:243-249) and bench.py's value_e2e.
"""
import numpy as np

from . import _ffi
from .dist import CocoRecordError, coco_record_layout, unpack_coco_records


class RecordPipeline:
    def __init__(self, net, batch, gather=None):
        """net: Yolact / MaskRCNN wrapper; batch: images per step (fixed: the block size depends on it; a short last batch is padded by the
        caller's bookkeeping, empty image slots simply carry count 0); gather: an isegmi.dist.RcclGather sized net.coco_record_bytes(batch)."""
        self.net, self.batch, self.gather = net, int(batch), gather
        self.kind = net.KIND
        self.K = int(net.cfg.max_num_detections if self.kind == 1 else net.cfg.det_cap)
        self.has_mscore = bool(self.kind == 1 and getattr(net, "has_maskiou", False))
        self.nbytes, self.chars_off = net.coco_record_bytes(self.batch)
        secs, coff, _ = coco_record_layout(self.batch, self.K, self.kind, self.has_mscore, 0)
        assert coff == self.chars_off, (coff, self.chars_off)
        self.cap_chars = self.nbytes - self.chars_off
        if gather is None:
            self.dev = [_ffi.DeviceBuffer((self.nbytes,), np.uint8) for _ in range(2)]
            self.pin = [_ffi.PinnedBuffer((self.nbytes,), np.uint8) for _ in range(2)]
        else:
            assert gather.nbytes == self.nbytes, (gather.nbytes, self.nbytes)
            self.dev = []
            # every rank's blocks come down asynchronously too, behind the all-gather on the results stream (a synchronous copy of world x
            # 2 MB into pageable memory per step would be host time the producer loop does not have at 2-3 ms per step)
            self.pin = [_ffi.PinnedBuffer((self.nbytes * gather.world,), np.uint8) for _ in range(2)]
        self.step = 0
        self.strict = True   # an RLE overflow raises CocoRecordError; run_record_loop clears it and redoes the step with larger capacities
        self._closed = False
        self.pending = []  # (slot, meta) of steps whose records have not been handed out yet
        # the mask planes are consumed by the run-length encoder alone, through the box windows the paste / mask assembly kernels leave next to
        # them: their zero background (242 MB per Yolact bs=8 step, 107 MB per Mask R-CNN image) is not written while the pipeline is open
        net.set_param("sparse_masks", 1.0)

    def _emit(self, slot):
        if self.gather is None:
            self.net.download_fence(slot)        # the slot's previous download has left the device buffer
            self.net.pack_coco_records(self.dev[slot], self.batch)
            self.net.download_async(slot, self.pin[slot], self.dev[slot], self.nbytes)
        else:
            self.gather.gather_coco_from(self.net, self.batch)
            self.net.download_async(slot, self.pin[slot], self.gather.recv, self.nbytes * self.gather.world)

    def _collect(self, slot):
        """-> list over ranks of unpacked record dicts (one entry when there is no gather)."""
        self.net.download_wait(slot)
        if self.gather is None:
            return [self._unpack(self.pin[slot].array)]
        blocks = self.pin[slot].array.reshape(self.gather.world, self.nbytes)
        return [self._unpack(blocks[r]) for r in range(self.gather.world)]

    def _unpack(self, buf):
        r = unpack_coco_records(buf, self.batch, self.K, self.kind, self.has_mscore, self.cap_chars, strict=self.strict)
        return {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in r.items()}  # copies: the pinned slot is reused two steps later

    def submit(self, meta=None):
        """Call after net.rle_device() of the current step.  Returns (meta, [records per rank]) of the previous step, or None."""
        slot = self.step % 2
        out = None
        if len(self.pending) == 2:  # the slot about to be reused still holds an unread step
            out = self._pop()
        self._emit(slot)
        self.pending.append((slot, meta))
        self.step += 1
        if out is None and len(self.pending) == 2:
            out = self._pop()
        return out

    def submit_empty(self, meta=None):
        """A step without a batch on this rank (several ranks, image list not divisible): contributes an all-zero block to the all-gather."""
        assert self.gather is not None, "only the multi-rank pipeline has collective steps"
        slot = self.step % 2
        self.gather.gather_empty(self.net)              # on the results stream, like every data step: one stream per communicator
        self.gather.fence_results_stream(self.net)      # the copy below is ordered behind it
        self.net.download_async(slot, self.pin[slot], self.gather.recv, self.nbytes * self.gather.world)
        self.pending.append((slot, meta))
        self.step += 1
        return self._pop() if len(self.pending) == 2 else None

    def _pop(self):
        slot, meta = self.pending.pop(0)
        return meta, self._collect(slot)

    def flush(self):
        """Records of the steps still in flight, oldest first."""
        out = []
        while self.pending:
            out.append(self._pop())
        return out

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        """Idempotent, and safe on the way out of an exception: whatever happened, the engine writes whole mask planes again afterwards (a
        compute_prediction() on the same engine would otherwise return masks with an uninitialised background) and the buffers are freed."""
        if self._closed:
            return
        self._closed = True
        try:
            self.net.sync()
        finally:
            try:
                self.net.set_param("sparse_masks", 0.0)
            finally:
                for b in self.dev + self.pin:
                    try:
                        b.free()
                    except Exception:
                        pass
                self.dev, self.pin, self.pending = [], [], []


def schedule_batches(group_keys, batch_size):
    """maskrcnn-benchmark's GroupedBatchSampler over a sequential sampler (the test loader, README.md:344-347): images with the same
    group key form batches of `batch_size` in data-set order (the last batch of a group may be short), and the batches are then ordered
    by their first image.  -> list of lists of image indices."""
    groups = {}
    for i, k in enumerate(group_keys):
        groups.setdefault(k, []).append(i)
    batches = [idx[j:j + batch_size] for idx in groups.values() for j in range(0, len(idx), batch_size)]
    batches.sort(key=lambda b: b[0])
    return batches


def record_capacity(net):
    """K: detection slots per image in the record block."""
    return int(net.cfg.max_num_detections if net.KIND == 1 else net.cfg.det_cap)


def grow_rle_capacity(net, overflows, batch):
    """Raises the engine's RLE capacities after a step overflowed them.  `overflows`: the (bits, runs, characters) of every rank's block of
    that step -- all ranks see all of them (they are in the gathered blocks), so all ranks arrive at the same new capacities."""
    nb, co = net.coco_record_bytes(batch)
    runs, chars = max(o[1] for o in overflows), max(o[2] for o in overflows)
    new_chars = max(2 * (nb - co), int(chars * 1.25) + 4096, 3 * runs)   # ~1.5-2 characters per run; a second round settles it if not
    net.set_param("rle_cap_chars", float(new_chars))
    if any(o[0] & 1 for o in overflows):
        net.set_param("rle_cap_runs", float(max(2 * runs, int(runs * 1.25) + 4096)))
    return new_chars


def run_record_loop(net, batch, nsteps, enqueue, consume, rank=0, world=1, force_gather=False, max_regrow=6):
    """The step loop of the device-side COCO output, shared by predictor.inference (Mask R-CNN) and yolact.evaluate:
        enqueue(step, slot) -> True after enqueueing upload + forward + masks + rle_device of this rank's batch of `step` (slot: which of the two
                               pinned / input buffers to use), False if this rank has no batch in that step (it then contributes an empty block);
        consume(step, [records per rank]) receives every step's records exactly once (in step order, except that a redone step comes after
                               the step that was already in flight behind it).
    One RecordPipeline (and one RCCL gather for several ranks) lives for the loop and is closed on EVERY way out, exceptions included.
    A step whose run lengths overflow the engine's RLE capacities (unpack_coco_records: status bits) is not fatal: the pipeline is drained,
    the capacities are raised (every rank sees every rank's status, so all ranks agree), the block and the gather buffers are re-sized and the
    step is enqueued again -- its images are still with the caller -- before the loop goes on (one overflow used to abort a whole
    data set, and left the other ranks waiting in the all-gather)."""
    gather = make_gather(net, batch, rank, world, force_gather)
    pipe = None
    try:
        pipe = RecordPipeline(net, batch, gather)
        pipe.strict = False
        redo = []

        def handle(done):
            step, recs = done
            ov = [r["overflow"] for r in recs if r["overflow"] is not None]
            if ov:
                redo.append((step, ov))
            else:
                consume(step, recs)

        def submit(step, slot):
            has = enqueue(step, slot)
            return pipe.submit(step) if has else pipe.submit_empty(step)

        def settle():
            """drain, grow, redo (in step order) until every step that overflowed has fitted"""
            nonlocal pipe
            rounds = 0
            while redo:
                for d in pipe.flush():
                    handle(d)
                redo.sort(key=lambda t: t[0])
                todo, ovs = [t[0] for t in redo], [o for t in redo for o in t[1]]
                del redo[:]
                rounds += 1
                if rounds > max_regrow:
                    raise CocoRecordError("device RLE capacities still overflow after %d rounds of growth (step %d)" % (max_regrow, todo[0]))
                pipe.close()
                grow_rle_capacity(net, ovs, batch)
                if gather is not None:
                    gather.resize(net.coco_record_bytes(batch)[0])
                pipe = RecordPipeline(net, batch, gather)
                pipe.strict = False
                for s_ in todo:
                    d = submit(s_, s_ & 1)
                    if d is not None:
                        handle(d)
                for d in pipe.flush():
                    handle(d)

        for step in range(nsteps):
            done = submit(step, step & 1)
            if done is not None:
                handle(done)
            settle()
        for d in pipe.flush():
            handle(d)
        settle()
        net.sync()
    finally:
        try:
            if pipe is not None:
                pipe.close()
        finally:
            if gather is not None:
                from .dist import rendezvous_cleanup
                try:
                    gather.close()
                finally:
                    rendezvous_cleanup(rank, world)


def make_gather(net, batch, rank, world, force=False):
    """The RCCL all-gather of a multi-rank run (None for one rank unless `force`: a world of one still goes through RCCL, for tests on a
    one-GPU box): unique id from rank 0 through the stdlib rendezvous of isegmi.dist."""
    if world <= 1 and not force:
        return None
    from .dist import RcclGather, rendezvous_unique_id
    uid = rendezvous_unique_id(rank, world, RcclGather.unique_id)
    return RcclGather(rank, world, uid, net.coco_record_bytes(batch)[0])
