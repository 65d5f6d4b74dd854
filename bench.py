#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RoI/mask inference hot path.

`python bench.py --gpus N --steps K --warmup W`
  N = 1: runs in this process.  N > 1 without a rank environment: this process only starts N child ranks (`python -m isegmi.launch
  --nproc N bench.py ...`, before any HIP call), relays rank 0's JSON line and exits with the children's code.  N > 1 inside a rank
  environment (RANK / WORLD_SIZE set, by isegmi.launch or by the driver's launcher): one rank per GPU.

A "step" = one pass of the hot path over one batch of synthetic input, as SURVEY.md 8(d) defines the metric: the H2D of the input is
INSIDE the step (the uint8 images go up through pinned memory on a copy stream, double-buffered; FastBaseTransform / build_transform +
to_image_list run on the device), image decode / resize is outside.  `value_resident` repeats the K steps with the batch already in HBM,
`value_e2e` with everything a user of inference() / evaluate() gets: uint8 upload -> forward -> masks -> RLE on the device -> one
record block per step back in pinned host memory (N = 1) or all-gathered over RCCL (N > 1).
Default workload = BASELINE.json configs[1]: Yolact R50-FPN 550x550, bs=8 per GPU: backbone -> FPN -> protonet +
prediction heads -> Detect (softmax, decode, fast-NMS) -> postprocess (mask assembly at 550x550, uint8)
[-> RCCL all-gather of detection records when N>1].  At N=1 the default run ALSO measures BASELINE configs[2] /
north_star's own target (Mask R-CNN R50-FPN 1333x800: bs=2 throughput, bs=1 latency, conv roofline fraction, parity
against the CPU oracle on the bench batch) and reports it under the "maskrcnn" key of the same JSON line.
`--model maskrcnn [--depth 101] [--fp16] [--c4]` makes Mask R-CNN the measured workload itself.

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "instancesegmentation-jittor_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak (spec)
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16 MFMA peak (spec, no sparsity)
PROFILE_ROUND = "r06"
ALLOW_STALE_TRAFFIC = False   # --allow-stale-traffic


def pmc_traffic(name):
    """HBM bytes per conv launch from THIS round's committed PMC summary (tools/pmc_summary.py), or None.  A summary of an older round
    belongs to another binary: it is refused (traffic null, the note says so) unless --allow-stale-traffic, and then the note names it."""
    for rnd in ((PROFILE_ROUND, "r04", "r03", "r02", "r01") if ALLOW_STALE_TRAFFIC else (PROFILE_ROUND,)):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_%s" % (rnd, name))) as f:
                return json.load(f)["conv_mfma_kernel_all"]["hbm_bytes_per_launch"], "%s_%s%s" % (rnd, name, "" if rnd == PROFILE_ROUND else " (STALE: an older round's binary)")
        except Exception:
            continue
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--spawn-timeout", type=float, default=900.0, help="--gpus N > 1 started from a plain shell: seconds before the child ranks are killed")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU per step (default 8 yolact, 2 maskrcnn)")
    ap.add_argument("--model", default="yolact", choices=["yolact", "maskrcnn"])
    ap.add_argument("--depth", type=int, default=50, choices=[50, 101], help="maskrcnn: ResNet depth")
    ap.add_argument("--fp16", action="store_true", help="fp16 storage + f16 MFMA convs: maskrcnn = BASELINE configs[4]; yolact = optional mode (the headline configs[1] is fp32: default)")
    ap.add_argument("--c4", action="store_true", help="maskrcnn: the R-50-C4 config (README.md:263-273) instead of R-50/101-FPN")
    ap.add_argument("--yolact-config", default="resnet50", choices=["resnet50", "base", "im700", "plus_resnet50", "plus_base", "darknet53"],
                    help="yolact: which upstream config (the headline configs[1] is resnet50: default); plus_* = YOLACT++ (DCNv2 backbone, 9 anchors, mask re-scoring)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8, help="images the CPU oracle is timed on")
    ap.add_argument("--no-latency", action="store_true", help="skip the extra bs=1 latency pass")
    ap.add_argument("--no-maskrcnn", action="store_true", help="default yolact run: skip the extra Mask R-CNN measurements (R50-FPN fp32 bs=2 = configs[2]; R101-FPN fp16 bs=8 = configs[4])")
    ap.add_argument("--no-r101f16", action="store_true", help="default yolact run: skip the extra Mask R-CNN R101-FPN fp16 bs=8 measurement (configs[4] per-GPU shape)")
    ap.add_argument("--no-box", action="store_true", help="skip the ~0.1 s box calibration (bare MFMA / copy loops)")
    ap.add_argument("--allow-stale-traffic", action="store_true", help="roofline.traffic may come from an older round's committed PMC summary")
    ap.add_argument("--f16-mfma-shape", type=int, default=-1, choices=[-1, 0, 1, 2, 3], help="A/B: isegmi_set_f16_mfma_shape (0: 32x32x16 everywhere, 1: row strips on 16x16x32, 2: persistent tiles too, 3 (library default): 1 + the 144-row tiles)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the extra timed loops (batch resident in HBM; fp32 upload)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end loop (uint8 upload -> ... -> RLE -> record block on the host)")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE", help="engine parameter for A/B runs (isegmi_engine_set_param), repeatable")
    ap.add_argument("--single-stream", action="store_true", help="profiling aid: run the timed region on one stream too, so that "
                    "rocprofv3 per-kernel durations are not inflated by overlapping launches (throughput drops ~20 %%)")
    return ap.parse_args()


_JSON_FD = None


def emit(line):
    """The ONE JSON line goes to the process's original stdout; fd 1 itself points at stderr for the whole run, so
    that library banners (RCCL version banner, gloo's connection message ...) cannot precede or follow it."""
    os.write(_JSON_FD if _JSON_FD is not None else 1, (line + "\n").encode())


def spawn_ranks(a):
    """--gpus N > 1 from a plain shell: start the N ranks as CHILD processes with the package's own launcher (`python -m isegmi.launch`:
    fresh interpreters, RANK / WORLD_SIZE / MASTER_* environment, a watchdog on communicator creation; nothing in this process has touched
    HIP), relay rank 0's JSON line, propagate the exit code."""
    cmd = [sys.executable, "-m", "isegmi.launch", "--nproc", str(a.gpus), "--timeout", str(a.spawn_timeout), "--init-timeout", "300",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=a.spawn_timeout + 30)
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
        sys.stderr.write("bench.py: the %d-rank run did not finish within %.0f s; its process group was killed\n" % (a.gpus, a.spawn_timeout))
        raise SystemExit(124)
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank run failed (rc %d)\n" % (a.gpus, proc.returncode))
        raise SystemExit(proc.returncode or 1)
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    raise SystemExit(0)


def set_omp_threads(n):
    """The oracle's OpenMP team size (libgomp is loaded with liboracle.so); OMP_NUM_THREADS only acts before that load."""
    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(q * len(xs)))] if xs else None


class Dist:
    """The N > 1 harness of the driver contract (one rank per GPU; barrier and MAX over ranks around the timed regions) on the product's own
    transport: the ranks -- started by `python -m torch.distributed.run` (the driver) or by `python -m isegmi.launch` (spawn_ranks), both set
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* -- meet through isegmi.dist.rendezvous_unique_id ONCE and share ONE RCCL communicator per rank
    for everything: the 8-byte control words (barrier, MAX), the detection-record blocks of the timed loops and the COCO record blocks of
    the end-to-end loop (`make_gather` re-sizes the same isegmi.dist.RcclGather; round 4 built three communicators per rank -- three
    ncclCommInitRank calls to hang in).  A control word goes out only on an idle communicator (RcclGather.allgather_bytes waits for this
    rank's earlier gathers first), so the issue order of collectives is the program order on every rank.  Nothing here imports torch."""

    def __init__(self, a):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.force = os.environ.get("ISEGMI_BENCH_FORCE_DIST") == "1"  # exercise the N>1 code path on a 1-GPU box
        if a.gpus != self.world:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (a.gpus, self.world))
            raise SystemExit(2)
        self.on = self.world > 1 or self.force
        self._g = None
        self.rdzv_seconds = None
        self.ranks = None

    def _gather(self):
        if self._g is None:
            from isegmi import _ffi
            from isegmi.dist import RcclGather, rendezvous_unique_id
            _ffi.set_device(self.local_rank)
            t0 = time.perf_counter()
            uid = rendezvous_unique_id(self.rank, self.world, RcclGather.unique_id)
            self.rdzv_seconds = time.perf_counter() - t0
            self._g = RcclGather(self.rank, self.world, uid, 4096)
            # who is there: every rank's (rank, device, seconds in ncclCommInitRank, seconds in the rendezvous) -- the first collective of the run
            mine = np.array([self.rank, self.local_rank, self._g.init_seconds, self.rdzv_seconds], np.float64)
            rows = self._g.allgather_bytes(mine.tobytes()).view(np.float64).reshape(self.world, 4)
            self.ranks = [{"rank": int(r[0]), "device_id": int(r[1]), "comm_init_s": round(float(r[2]), 3), "rendezvous_s": round(float(r[3]), 3)} for r in rows]
        return self._g

    def barrier(self):
        if self.on:
            self._gather().allgather_bytes(b"\0" * 8)

    def max(self, v):
        if not self.on:
            return v
        blocks = self._gather().allgather_bytes(np.array([v], np.float64).tobytes())
        return float(blocks.view(np.float64).max())

    def make_gather(self, nbytes):
        """The run's one RcclGather, sized for blocks of `nbytes` from here on (None at N = 1)."""
        if not self.on:
            return None
        g = self._gather()
        g.resize(nbytes)
        return g

    def info(self):
        return {"rccl_communicators_per_rank": 1, "ranks": self.ranks} if self.on and self.ranks else None

    def close(self):
        if self._g is not None:
            self.barrier()
            self._g.close()
            self._g = None


def timed_region(net, step, full_sync, dist, steps):
    """EXACTLY `steps` steps bracketed by barrier + full synchronisation on both sides; MAX over ranks.  Every step leaves a
    completion mark on the stream its results finish on: the intervals are per-step latency samples of the pipelined run."""
    dist.barrier()
    full_sync()
    net.step_times()  # drop stale marks
    net.mark_step()
    # The producer keeps at most two steps in flight: after enqueueing step i it waits (on a completion EVENT, not a device sync) for step
    # i - 1, as any loop that consumes its results does (isegmi.pipeline.RecordPipeline collects step i - 1's records at that point).  An
    # unbounded loop lets the host run ~15 steps ahead until the hardware queues fill; while they fill, every step of the upload pipeline
    # takes 0.1-0.2 ms longer than the one before (9.2 -> 11.1 ms over 17 steps, then back to 8.3: gpurun_out r3 experiments, DESIGN.md
    # section 5) -- a start-up transient of the queueing, not of the kernels.  ISEGMI_BENCH_DEPTH=0 restores the unbounded loop.
    depth = int(os.environ.get("ISEGMI_BENCH_DEPTH", "1"))
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
        net.mark_step()
        if depth > 0:
            net.wait_mark(depth)
    full_sync()
    dist.barrier()
    elapsed = dist.max(time.perf_counter() - t0)
    st = net.step_times()
    if os.environ.get("ISEGMI_BENCH_DUMP_STEPS") == "1":  # diagnostics: every step's completion interval
        sys.stderr.write("step intervals (ms): " + " ".join("%.2f" % v for v in st) + "\n")
    return elapsed, st


PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec; ~6.3 TB/s measured with a float4 copy)


def op_stats(net):
    """(label, us, algorithmic bytes, timed scopes) of the HBM-bound stages accumulated under `op_timing` since the last call."""
    from isegmi import _ffi
    cap = 64
    names = C.create_string_buffer(16384)
    us, by, ln, cnt = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_int64 * cap)(), C.c_int()
    _ffi.check(_ffi.lib().isegmi_engine_op_stats(net._h, names, 16384, us, by, ln, cap, C.byref(cnt)))
    labels = names.value.decode().split("\n") if cnt.value else []
    return [(labels[i], float(us[i]), float(by[i]), int(ln[i])) for i in range(cnt.value)]


def hbm_rooflines(ops, steps, traffic=None):
    """SURVEY 8(d) "Bounding roofline per stage": the HBM-bound stages as {kernel, bytes, us, frac} per step -- achieved = algorithmic bytes /
    HIP-event time, against the 8 TB/s HBM peak; `traffic` = counter bytes (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE) from the committed PMC pass."""
    out = []
    for label, us, by, ln in sorted(ops, key=lambda t: -t[1]):
        if us <= 0:
            continue
        gbs = by / us / 1e3 if by > 0 else None
        out.append({"kernel": label, "bytes": int(by / max(steps, 1)), "us": round(us / max(steps, 1), 2), "launches": ln // max(steps, 1),
                    "achieved": None if gbs is None else round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None if gbs is None else round(gbs / PEAK_HBM_GBS, 4),
                    "traffic": (traffic or {}).get(label)})
    return out


def roofline_pass(net, step, full_sync, steps, single_stream):
    """The same K steps again, single-stream, every conv launch bracketed by HIP events on the engine stream (per-launch
    durations are not meaningful while launches from several streams overlap); the HBM-bound stages are bracketed the same way.
    -> (conv FLOPs, conv ms, conv launches, [(stage label, us, algorithmic bytes, scopes)])"""
    from isegmi import _ffi
    f, m, l = C.c_double(), C.c_double(), C.c_int64()
    net.set_param("multi_stream", 0.0)
    net.set_param("conv_timing", 1.0)
    net.set_param("op_timing", 1.0)
    _ffi.check(_ffi.lib().isegmi_engine_conv_stats(net._h, C.byref(f), C.byref(m), C.byref(l)))  # reset
    op_stats(net)
    for i in range(steps):
        step(i)
    full_sync()
    _ffi.check(_ffi.lib().isegmi_engine_conv_stats(net._h, C.byref(f), C.byref(m), C.byref(l)))
    ops = op_stats(net)
    net.set_param("conv_timing", 0.0)
    net.set_param("op_timing", 0.0)
    net.set_param("multi_stream", 0.0 if single_stream else 1.0)
    return f.value, m.value, l.value, ops


def rle_roofline(net, run_once, full_sync, batch, K, single_stream, reps=5):
    """The RLE chain (csrc/rle.hip) as an HBM-bound stage: HIP-event time of the seven launches; algorithmic bytes = every valid mask's box
    WINDOW of its uint8 plane read once (1 B per pixel; the windows are data: taken from det.mask_window after the pass) + 8 B per 64 window
    pixels of run-start words written and read back."""
    net.set_param("multi_stream", 0.0)   # (one stream: a stage's events must not bracket another stream's kernels)
    net.set_param("op_timing", 1.0)
    op_stats(net)
    for _ in range(reps):
        run_once()
    full_sync()
    allops = op_stats(net)
    ops = [o for o in allops if o[0].startswith("rle")]
    front = [o for o in allops if o[0].startswith("front end")]
    net.set_param("op_timing", 0.0)
    net.set_param("multi_stream", 0.0 if single_stream else 1.0)
    if not ops:
        return []
    win, cn = net.fetch("det.mask_window", batch).reshape(batch, -1, 4).astype(np.int64), net.fetch("det.count", batch)
    px = sum(int(max(0, w[2] - w[0]) * max(0, w[3] - w[1])) for i in range(batch) for w in win[i, :int(cn[i])])
    label, us, _, ln = ops[0]
    return [(label, us, (px + px / 64.0 * 16.0) * reps, ln)] + front


def hbm_traffic(name):
    """counter bytes per HBM-bound stage from the committed PMC summary (tools/hbm_stage_traffic.py), or None"""
    try:
        with open(os.path.join(ROOT, "profiles", "%s_%s" % (PROFILE_ROUND, name))) as f:
            return json.load(f)
    except Exception:
        return None


def e2e_region(net, pipe, upload, run_rle, full_sync, dist, steps):
    """K steps of the product path: uint8 upload (pinned, copy stream) -> forward -> masks -> RLE -> record block -> pinned host memory
    (or RCCL all-gather), the host collecting step i-1's block while step i runs (isegmi.pipeline.RecordPipeline).  Returns
    (elapsed, number of record blocks collected, total RLE characters, host seconds spent unpacking)."""
    upload(0)
    dist.barrier(); full_sync()
    blocks = chars = 0
    t0 = time.perf_counter()
    for i in range(steps):
        upload((i + 1) & 1)
        run_rle(i & 1)
        done = pipe.submit(i)
        if done is not None:
            blocks += 1; chars += sum(len(r["chars"]) for r in done[1])
    for done in pipe.flush():
        blocks += 1; chars += sum(len(r["chars"]) for r in done[1])
    full_sync(); dist.barrier()
    return dist.max(time.perf_counter() - t0), blocks, chars


def latency_pass(net, run1, iters=13, drop=3):
    lat = []
    for i in range(iters):
        net.sync()
        ts = time.perf_counter()
        run1()
        net.sync()
        lat.append((time.perf_counter() - ts) * 1e3)
    lat = sorted(lat[drop:])
    return {"p50_ms_per_image": round(lat[len(lat) // 2], 3), "img_per_s": round(1e3 / lat[len(lat) // 2], 2)}


STREAM_ROLES = ("main", "side0", "side1", "side2", "tail", "heads", "heads_side0", "heads_side1", "heads_side2", "copy")


def stream_layout(net):
    """Which of the engine's ten streams share an in-order hardware queue (isegmi_engine_stream_layout: equal numbers = one queue).  The
    runtime folds a process's streams onto four queues by its stream-creation history; the engine deals its roles over probed candidates
     -- printed so that a slow line can be told from a bad placement."""
    from isegmi import _ffi
    q = (C.c_int32 * 10)()
    net.sync()
    _ffi.check(_ffi.lib().isegmi_engine_stream_layout(net._h, q, 10))
    cls = [int(v) for v in q]
    ok = cls[0] not in (cls[4], cls[5], cls[9], cls[1]) and cls[4] not in (cls[5], cls[9])
    return {"queue_class": dict(zip(STREAM_ROLES, cls)), "designed_partition": bool(ok),
            "note": "main's queue carries neither tail, heads, copy nor side0; the tail's queue carries neither heads nor copy"}


def box_calibration():
    """Bare loops on this box right before the engines are built (~0.1 s): what the MFMA pipes and HBM of THIS device sustain."""
    from isegmi import _ffi
    b = _ffi.box_calibrate(30.0)
    b["note"] = ("~30 ms each: dependent v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_f16 chains on random register operands, four waves per SIMD, no memory traffic "
                 "(nominal 157.3 / 2516 TF/s at 2.4 GHz: the quotient is the clock the box holds); float4 copy of 1 GiB, read + written bytes "
                 "(a figure to compare boxes with: 4.7-4.9 TB/s on the boxes seen so far; a tuned copy reaches ~6.3 on this chip)")
    return b


def roofline_dict(kernel, flops, ms, launches, steps, peak, traffic_file):
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    traffic, src = pmc_traffic(traffic_file) if traffic_file else (None, None)
    return {"bound": "mfma", "kernel": kernel,
            "pass": "K single-stream steps right after the timed region, HIP events around every conv launch on the engine stream",
            "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
            "traffic_note": ("HBM bytes per conv launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (profiles/%s; FETCH x2 gfx950 correction); not collected live" % src) if traffic else
                            "null: no committed PMC pass of this round's binary for this configuration (an older round's summary is refused unless --allow-stale-traffic)",
            "algorithmic_gflop_per_step": round(flops / max(steps, 1) / 1e9, 2), "conv_ms_per_step": round(ms / max(steps, 1), 3),
            "conv_ms_note": "single-stream figure: in the pipelined multi-stream timed region launches of two steps overlap, so conv_ms_per_step may exceed ms_per_step",
            "launches_per_step": launches // max(steps, 1), "avg_launch_us": round(ms * 1e3 / max(launches, 1), 2)}


# ------------------------------------------------------------------------------------------------------------------ Yolact
def bench_yolact(a, dist):
    from isegmi import _ffi
    from isegmi.dist import record_bytes
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, YolactConfig, fast_base_transform
    a.batch = a.batch or 8
    rank, world = dist.rank, dist.world
    ycfg = {"resnet50": YolactConfig(), "base": YolactConfig.base(), "im700": YolactConfig.im700(),
            "plus_resnet50": YolactConfig.plus_resnet50(), "plus_base": YolactConfig.plus_base(), "darknet53": YolactConfig.darknet53()}[a.yolact_config]
    yname = {"resnet50": "Yolact R50-FPN", "base": "Yolact R101-FPN (yolact_base)", "im700": "Yolact R101-FPN 700 (yolact_im700)",
             "plus_resnet50": "YOLACT++ R50-FPN (DCNv2, 9 anchors, mask re-scoring)", "plus_base": "YOLACT++ R101-FPN (DCNv2 every 3rd block, 9 anchors, mask re-scoring)",
             "darknet53": "Yolact Darknet53-FPN (yolact_darknet53)"}[a.yolact_config]
    sd = yolact_state_dict(1234, ycfg.depth, ycfg.num_priors, ycfg.dcn_layers, ycfg.dcn_interval, ycfg.use_maskiou, ycfg.backbone)
    net = Yolact(sd, ycfg, max_batch=a.batch, device=dist.local_rank, fp16=a.fp16)
    ypeak = PEAK_F16_MFMA_TFLOPS if a.fp16 else PEAK_F32_MFMA_TFLOPS
    if a.single_stream:
        net.set_param("multi_stream", 0.0)
    for kv in a.param:
        net.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    size = net.size
    rng = np.random.default_rng(20261003 + rank)
    raw_u8 = rng.integers(0, 256, (a.batch, size, size, 3), dtype=np.uint8)  # what cv2.imread hands FastBaseTransform
    raw = raw_u8.astype(np.float32)
    if ycfg.backbone == "darknet53":
        from isegmi.yolact import darknet_base_transform
        imgs = darknet_base_transform(raw)
    else:
        imgs = fast_base_transform(raw)
    net.upload(imgs)
    gather = dist.make_gather(record_bytes(a.batch))
    pin8 = _ffi.PinnedBuffer(raw_u8.shape, np.uint8)
    pin8.array[...] = raw_u8

    def run(slot=0):
        net.forward_device(a.batch, slot)
        net.postprocess_device(size, size)
        if gather is not None:
            gather.gather_from(net)

    def upload_u8(slot):  # what cv2.imread hands FastBaseTransform: the uint8 images cross PCIe, the transform runs on the device
        net.upload_u8_async(pin8, a.batch, size, size, slot)

    def step(i):          # H2D of the input inside the step; batch i+1 goes up on the copy stream while batch i computes
        upload_u8((i + 1) & 1)
        run(i & 1)

    def step_resident(i):
        run(0)

    def full_sync():
        net.sync()
        if gather is not None:
            gather.wait()
        _ffi.sync()

    upload_u8(0)
    for i in range(max(a.warmup, 1 if gather is not None else 0)):
        step(i)
    full_sync()
    upload_u8(0)
    elapsed, step_ms = timed_region(net, step, full_sync, dist, a.steps)
    # what the GPU produced for the bench batch through the timed path (device front end + forward + postprocess); compared with the oracle below
    net.upload(imgs)
    conv_flops, conv_ms, conv_launches, ops = roofline_pass(net, step_resident, full_sync, a.steps, a.single_stream)
    gpu = {k: net.fetch(k, a.batch) for k in ("det.count", "det.prior", "det.class", "det.score", "det.box", "det.coeff", "det.masks", "det.box_int")}
    rccl = None
    if gather is not None:
        from isegmi.dist import unpack_records
        blocks = gather.fetch()
        rccl = {"rccl_ranks": int(gather.world), "ranks_with_records": int(sum(1 for r in range(world) if unpack_records(blocks[r], a.batch)["count"].any()))}
    resident_elapsed = h2d_elapsed = e2e = rle_op = None
    if not a.no_h2d:
        resident_elapsed, _ = timed_region(net, step_resident, full_sync, dist, a.steps)
        pinned = _ffi.PinnedBuffer(imgs.shape)
        pinned.array[...] = imgs

        def step_f32(i):
            net.upload_async(pinned, (i + 1) & 1)
            run(i & 1)
        net.upload_async(pinned, 0)
        h2d_elapsed, _ = timed_region(net, step_f32, full_sync, dist, a.steps)
        full_sync()
        pinned.free()
    if not a.no_e2e:
        from isegmi.pipeline import RecordPipeline
        cgather = dist.make_gather(net.coco_record_bytes(a.batch)[0]) if gather is not None else None
        def run_rle(slot):
            net.forward_device(a.batch, slot)
            net.postprocess_device(size, size)
            net.rle_device()
        e2e_sync = (lambda: (full_sync(), cgather.wait())) if cgather is not None else full_sync
        with RecordPipeline(net, a.batch, cgather) as pipe:   # closed on every way out: full mask planes again afterwards
            e2e_region(net, pipe, upload_u8, run_rle, e2e_sync, dist, max(2, a.warmup // 2))
            e2e_elapsed, blocks, chars = e2e_region(net, pipe, upload_u8, run_rle, e2e_sync, dist, a.steps)
            e2e = {"elapsed": e2e_elapsed, "blocks": blocks, "chars_per_step": chars / max(blocks, 1), "record_bytes": pipe.nbytes}
            K = pipe.K
        # the strings the device made for the bench batch, against the host encoder on the uint8 planes (rank 0, after the timed loops)
        net.upload(imgs)
        rle_op = rle_roofline(net, lambda: (upload_u8(0), net.forward_device(a.batch), net.postprocess_device(size, size), net.rle_device()), full_sync, a.batch, K, a.single_stream)
        from isegmi.coco import rle_counts, rle_to_string
        so, ch = net.fetch("rle.str_off"), net.fetch("rle.chars").tobytes()
        mk, cn = net.fetch("det.masks", a.batch), net.fetch("det.count", a.batch)
        e2e["rle_checked"] = 0
        e2e["rle_ok"] = True
        for i in range(a.batch):
            for k in range(0, int(cn[i]), 7):  # every 7th detection: the host encoder takes ~3 ms per 550x550 mask
                e2e["rle_ok"] &= ch[so[i * K + k]:so[i * K + k + 1]].decode() == rle_to_string(rle_counts(mk[i, k]))
                e2e["rle_checked"] += 1
    net.upload(imgs)
    full_sync()
    pin8.free()
    counts = gpu["det.count"]
    value = a.batch * world * a.steps / elapsed
    if rank != 0:
        return None, net, gather
    fp16_note = "fp16 storage / f16 MFMA, fp32 accumulate (optional mode, not configs[1])" if a.fp16 else "fp32"
    out = {
        "metric": "images/sec (%s %dx%d, bs=%d per GPU, %s)" % (yname, size, size, a.batch, fp16_note),
        "value": round(value, 2), "unit": "img/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if a.fp16 else "f32", "data": "synthetic",
        "config": {"workload": "%s %dx%d bs=%d/GPU random weights: uint8 upload + FastBaseTransform on the device + backbone+FPN+protonet+heads+Detect(fast-NMS)+%dx%d mask assembly%s" % (
                       yname, size, size, a.batch, size, size, " (BASELINE configs[1])" if a.yolact_config == "resnet50" and not a.fp16 else " (variant, not configs[1])"),
                   "global_batch": a.batch * world, "parallelism": "batch-sharded x%d, RCCL all-gather of detections" % world,
                   "detections_per_image_rank0": [int(c) for c in counts]},
        "value_note": "SURVEY 8(d): the H2D of every batch is inside the step (%.1f MB of uint8 images per batch through pinned memory on a copy stream, double-buffered; the transform runs on the engine's stream); results stay on the device (value_e2e ships them)" % (raw_u8.nbytes / 1e6),
        "roofline": roofline_dict(("conv_f16_glds / conv3x3_f16_strip kernels (all conv launches of a step, v_mfma_f32_32x32x16_f16)" if a.fp16 else
                                   "conv_mfma_v2_kernel + conv_mfma16_kernel + conv_mfma_kernel (all conv launches of a step; v_mfma_f32_32x32x2_f32 on 64x64 tiles, v_mfma_f32_16x16x4_f32 on the 32x32 / 32x64 blocks of small grids)"),
                                  conv_flops, conv_ms, conv_launches, a.steps, ypeak, None if (a.fp16 or a.yolact_config != "resnet50") else "pmc_yolact.json"),
        "step_ms": {"mean": round(elapsed / a.steps * 1e3, 3), "p50": round(pct(step_ms, 0.5), 3), "p90": round(pct(step_ms, 0.9), 3),
                    "note": "intervals between consecutive per-step completion events on the results stream (pipelined multi-stream run)" +
                            ("; the RCCL all-gather of the step runs on that stream in front of its mark, so it is inside these intervals" if gather is not None else "")},
        "mean_ms_per_image": round(elapsed / a.steps * 1e3 / a.batch, 3),
        "p50_ms_per_image": round(pct(step_ms, 0.5) / a.batch, 3),
    }
    if resident_elapsed is not None:
        out["value_resident"] = round(a.batch * world * a.steps / resident_elapsed, 2)
        out["value_incl_h2d_f32"] = round(a.batch * world * a.steps / h2d_elapsed, 2)
        out["h2d_note"] = "value_resident: the same K steps with the batch already in HBM; value_incl_h2d_f32: with the host-transformed fp32 batch (%.1f MB) uploaded every step instead of the uint8 images" % (imgs.nbytes / 1e6)
    if e2e is not None:
        out["value_e2e"] = round(a.batch * world * a.steps / e2e["elapsed"], 2)
        out["e2e"] = {"ms_per_step": round(e2e["elapsed"] / a.steps * 1e3, 3), "record_block_bytes": int(e2e["record_bytes"]),
                      "rle_chars_per_step": int(e2e["chars_per_step"]), "blocks_collected": int(e2e["blocks"]),
                      "device_rle_equals_host_encoder": bool(e2e["rle_ok"]), "masks_checked": int(e2e["rle_checked"]),
                      "note": "uint8 upload -> forward -> mask assembly -> pycocotools RLE on the device -> one fixed-size record block per step (boxes, scores, classes, RLE strings) "
                              "downloaded into pinned host memory%s while the next step runs; the %d MB of uint8 mask planes never leave the GPU" % (
                                  " / all-gathered over RCCL" if gather is not None else "", int(a.batch * 100 * size * size / 1e6))}
    if rccl:
        out.update(rccl)
    traffic = hbm_traffic("hbm_stage_traffic_yolact.json") if (a.yolact_config == "resnet50" and not a.fp16 and a.batch == 8) else None
    hbm = hbm_rooflines(ops, a.steps, traffic)
    if rle_op:
        hbm += hbm_rooflines(rle_op, 5, traffic)   # (the front end's counter bytes are per launch, as its algorithmic bytes are)
    out["roofline_hbm"] = hbm
    out["roofline_hbm_note"] = ("SURVEY 8(d) HBM-bound stages, per step of %d images, in the single-stream pass of `roofline`: achieved = ALGORITHMIC bytes (each "
                                "stage's compulsory reads + writes, SURVEY 8d) / HIP-event time on the launching stream; peak 8 TB/s (spec); frac well under 1 on a "
                                "latency-bound selection stage says its grid is small, not that it moves many bytes" % a.batch)
    if not a.no_latency:
        out["bs1"] = latency_pass(net, lambda: (net.forward_device(1), net.postprocess_device(size, size)))
        f1, m1, l1, _ = roofline_pass(net, lambda i: (net.forward_device(1), net.postprocess_device(size, size)), full_sync, 10, a.single_stream)
        out["bs1"]["roofline"] = {"bound": "mfma", "achieved": round(f1 / (m1 * 1e-3) / 1e12, 2) if m1 > 0 else 0.0, "peak": ypeak, "unit": "TFLOP/s",
                                  "frac": round(f1 / (m1 * 1e-3) / 1e12 / ypeak, 4) if m1 > 0 else 0.0, "conv_ms_per_image": round(m1 / 10, 3),
                                  "pass": "10 single-stream bs=1 steps, HIP events around every conv launch"}
        if not a.fp16:   # the opt-in latency numerics mode (fixed-tree split-K on the backbone's small-M / large-K layers): same pass, reported NEXT TO the default
            net.set_param("conv_split_k", 1.0)
            sk = latency_pass(net, lambda: (net.forward_device(1), net.postprocess_device(size, size)))
            net.set_param("conv_split_k", 0.0)
            out["bs1"]["conv_split_k"] = dict(sk, note="engine parameter conv_split_k = 1 (default 0): another fp32 association of the same sums, bit-exact against the "
                                                       "oracle run in the same mode (tests/test_split_k_gpu.py); bs1.p50_ms_per_image above is the DEFAULT numerics")
        net.set_param("timing", 1.0)
        net.forward_device(a.batch); net.postprocess_device(size, size); net.sync()
        out["stage_ms_bs%d" % a.batch] = {k: round(v, 3) for k, v in net.timings()}
        net.set_param("timing", 0.0)

    # ---- CPU baseline: the oracle restatement on the host cores (rank 0, N=1 only); doubles as the parity check of this batch
    if world == 1 and not a.no_cpu_baseline:
        from oracle.yolact_ref import YolactRef
        ncpu = min(len(os.sched_getaffinity(0)), 16)  # the GPU box's CPU share for one GPU
        set_omp_threads(ncpu)
        ref = YolactRef(sd, max_size=ycfg.max_size, scales=ycfg.pred_scales, depth=ycfg.depth, scales_per_level=ycfg.scales_per_level,
                        square=ycfg.use_square_anchors)
        k = max(1, min(a.cpu_sample, a.batch))
        tc = time.perf_counter()
        done, dets, posts = 0, None, None
        while True:  # bounded sample: whole passes over k images until ~10 s of CPU work (cap: 6 passes)
            d_ = ref.forward(imgs[:k])
            p_ = [YolactRef.postprocess(d, size, size) for d in d_]
            dets, posts = dets or d_, posts or p_
            done += k
            tcpu = time.perf_counter() - tc
            if tcpu >= 10.0 or done >= 6 * k:
                break
        out["cpu_baseline"] = {"value": round(done / tcpu, 4), "unit": "img/s", "cores": ncpu, "kind": "port",
                               "sample": "%d images (passes over %d images of the bench batch), oracle/ C+numpy restatement (AVX2 FMA + OpenMP, %d threads), %.1f s" % (done, k, ncpu, tcpu)}
        set_omp_threads(1)
        tc = time.perf_counter()
        d1 = ref.forward(imgs[:1])
        YolactRef.postprocess(d1[0], size, size)
        t1 = time.perf_counter() - tc
        set_omp_threads(ncpu)
        out["cpu_baseline"]["value_1thread"] = round(1.0 / t1, 4)
        out["cpu_baseline"]["sample_1thread"] = "1 image of the bench batch, 1 thread, %.1f s" % t1
        if not a.fp16:  # the fp16 mode is tolerance-parity (tests), not index-exact
            ok = True
            for i in range(k):
                r, (rc, rs, rb, rm) = dets[i], posts[i]
                c = len(r["score"])
                ok &= int(gpu["det.count"][i]) == c
                ok &= all(np.array_equal(gpu[g][i, :c], r[o]) for g, o in (("det.prior", "prior"), ("det.class", "cls"), ("det.score", "score"),
                                                                          ("det.box", "box"), ("det.coeff", "mask")))
                if not ycfg.use_maskiou:
                    ok &= np.array_equal(gpu["det.box_int"][i, :c], rb) and np.array_equal(gpu["det.masks"][i, :c], rm)
            out["parity_vs_oracle_on_bench_batch"] = bool(ok)
            out["parity_note"] = "%d images: detection count, prior index, class, score, box, coefficients, integer boxes and %dx%d masks all bit-equal" % (k, size, size)
    return out, net, gather


# ------------------------------------------------------------------------------------------------------------ Mask R-CNN
def bench_maskrcnn(a, dist, summary=None):
    """Mask R-CNN R50-FPN 1333x800 (padded 800x1344), bs=2 per GPU: BASELINE configs[2]; step = forward + Masker paste.
    summary: the compact records the default (Yolact) run embeds -- "r50": configs[2] under "maskrcnn"; "r101f16": the per-GPU shape of
    configs[4] (R101-FPN, fp16 storage / f16 MFMA, bs=8) under "maskrcnn_r101_fp16" (no CPU leg: its full-size parity test runs under -m gpu)."""
    from isegmi import _ffi
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig, prepare_images
    from isegmi.weights import maskrcnn_state_dict
    batch = {"r50": 2, "r101f16": 8}[summary] if summary else (a.batch or 2)
    depth, fp16, c4 = {"r50": (50, False, False), "r101f16": (101, True, False)}[summary] if summary else (a.depth, a.fp16, a.c4)
    rank, world = dist.rank, dist.world
    if c4:
        from isegmi.weights import maskrcnn_c4_state_dict
        assert not fp16 and depth == 50, "--c4 is R-50, fp32"
        sd, mcfg = maskrcnn_c4_state_dict(1234), MaskRCNNConfig.c4()
    else:
        sd, mcfg = maskrcnn_state_dict(1234, depth=depth), MaskRCNNConfig(depth=depth)
    rng = np.random.default_rng(20261003 + rank)
    imgs_u8 = [rng.integers(0, 256, (800, 1333, 3), dtype=np.uint8) for _ in range(batch)]  # what PIL's resize hands build_transform
    imgs = [im.astype(np.float32) for im in imgs_u8]
    x, hw = prepare_images(imgs, mcfg.SIZE_DIVISIBILITY)
    model = MaskRCNN(sd, x.shape[1], x.shape[2], cfg=mcfg, max_batch=batch, device=dist.local_rank, fp16=fp16)
    if a.single_stream:
        model.set_param("multi_stream", 0.0)
    for kv in a.param:
        model.set_param(kv.split("=")[0], float(kv.split("=")[1]))
    tag = "R50-C4" if c4 else "R%d-FPN" % depth
    prec = "fp16 storage / f16 MFMA, fp32 accumulate" if fp16 else "fp32"
    peak = PEAK_F16_MFMA_TFLOPS if fp16 else PEAK_F32_MFMA_TFLOPS
    model.upload(x, hw)
    gather = None
    if not summary:
        from isegmi.dist import maskrcnn_record_bytes
        gather = dist.make_gather(maskrcnn_record_bytes(batch, M=14 if c4 else 28))

    flat = np.concatenate([im.reshape(-1) for im in imgs_u8])
    pin8 = _ffi.PinnedBuffer(flat.shape, np.uint8)
    pin8.array[...] = flat

    def run(slot=0):
        model.forward_device(batch, slot)
        model.paste_device(800, 1333)
        if gather is not None:
            gather.gather_from(model)

    def upload_u8(slot):  # what PIL's resize hands build_transform: uint8 images over PCIe, mean subtraction + to_image_list padding on the device
        model.upload_u8_async(pin8, hw, slot)

    def step(i):          # H2D of the input inside the step
        upload_u8((i + 1) & 1)
        run(i & 1)

    def step_resident(i):
        run(0)

    def full_sync():
        model.sync()
        if gather is not None:
            gather.wait()
        _ffi.sync()

    steps, warmup = a.steps, a.warmup
    upload_u8(0)
    for i in range(max(warmup, 1 if gather is not None else 0)):
        step(i)
    full_sync()
    upload_u8(0)
    elapsed, step_ms = timed_region(model, step, full_sync, dist, steps)
    model.upload(x, hw)
    flops, ms, launches, ops = roofline_pass(model, step_resident, full_sync, steps, a.single_stream)
    names = ("det.count", "det.score", "det.label", "det.box", "det.mask14" if c4 else "det.mask28", "det.masks", "proposal_count", "proposals")
    gpu = {k: model.fetch(k, batch) for k in names}
    rccl = None
    if gather is not None:
        from isegmi.dist import unpack_maskrcnn_records
        blocks = gather.fetch()
        rccl = {"rccl_ranks": int(gather.world), "ranks_with_records": int(sum(1 for r in range(world) if unpack_maskrcnn_records(blocks[r], batch, M=14 if c4 else 28)["count"].any()))}
    resident_elapsed = h2d_elapsed = e2e = rle_op = None
    lean = summary == "r101f16"   # (the embedded configs[4] block keeps the default run short: value, roofline, bs1)
    if not a.no_h2d and not lean:
        resident_elapsed, _ = timed_region(model, step_resident, full_sync, dist, steps)
        pinned = _ffi.PinnedBuffer(x.shape)
        pinned.array[...] = x

        def step_f32(i):
            model.upload_async(pinned, (i + 1) & 1)
            run(i & 1)
        model.upload_async(pinned, 0)
        h2d_elapsed, _ = timed_region(model, step_f32, full_sync, dist, steps)
        full_sync()
        pinned.free()
    if not a.no_e2e and not lean:
        from isegmi.pipeline import RecordPipeline
        cgather = dist.make_gather(model.coco_record_bytes(batch)[0]) if gather is not None else None
        def run_rle(slot):
            model.forward_device(batch, slot)
            model.paste_device(800, 1333)
            model.rle_device()
        e2e_sync = (lambda: (full_sync(), cgather.wait())) if cgather is not None else full_sync
        with RecordPipeline(model, batch, cgather) as pipe:   # closed on every way out: full mask planes again afterwards
            e2e_region(model, pipe, upload_u8, run_rle, e2e_sync, dist, max(2, warmup // 2))
            e2e_elapsed, nblocks, chars = e2e_region(model, pipe, upload_u8, run_rle, e2e_sync, dist, steps)
            e2e = {"elapsed": e2e_elapsed, "blocks": nblocks, "chars_per_step": chars / max(nblocks, 1), "record_bytes": pipe.nbytes}
            K = pipe.K
        model.upload(x, hw)
        rle_op = rle_roofline(model, lambda: (upload_u8(0), model.forward_device(batch), model.paste_device(800, 1333), model.rle_device()), full_sync, batch, K, a.single_stream)
        from isegmi.coco import rle_counts, rle_to_string
        so, ch = model.fetch("rle.str_off"), model.fetch("rle.chars").tobytes()
        mk, cn = model.fetch("det.masks", batch), model.fetch("det.count", batch)
        e2e["rle_checked"], e2e["rle_ok"] = 0, True
        for i in range(batch):
            for k in range(0, int(cn[i]), 9):  # every 9th detection: the host encoder takes ~10 ms per 800x1333 mask
                e2e["rle_ok"] &= ch[so[i * K + k]:so[i * K + k + 1]].decode() == rle_to_string(rle_counts(mk[i, k]))
                e2e["rle_checked"] += 1
    model.upload(x, hw)
    full_sync()
    pin8.free()
    if rank != 0:
        return None, model, gather
    value = batch * world * steps / elapsed
    traffic_file = "pmc_r101f16.json" if (fp16 and depth == 101 and batch == 8 and not c4) else None if (fp16 or c4 or depth != 50 or batch != 2) else "pmc_maskrcnn.json"
    roof = roofline_dict("conv_f16_glds / conv3x3_f16_strip kernels (all conv launches of a step)" if fp16 else "conv_mfma_v2_kernel + conv_mfma16_kernel + conv_mfma_kernel (all conv launches of a step)",
                         flops, ms, launches, steps, peak, traffic_file)
    workload = ("Mask R-CNN %s 1333x800 (padded 800x1344) bs=%d/GPU random weights, %s: uint8 upload + conv1-4 + single-map RPN (6000 -> 1000) + RoIAlign + conv5 head + NMS + shared-extractor mask branch + paste (the README.md:263-273 config; not a BASELINE config)" % (tag, batch, prec)) if c4 else \
        "Mask R-CNN %s 1333x800 (padded 800x1344) bs=%d/GPU random weights, %s: uint8 upload + mean/pad on the device + backbone+FPN+RPN+RoIAlign+box head+NMS+mask head+paste (BASELINE configs[%d])" % (tag, batch, prec, 4 if fp16 else 2)
    out = {"metric": "images/sec (Mask R-CNN %s 1333x800, bs=%d per GPU, %s)" % (tag, batch, prec),
           "value": round(value, 3), "unit": "img/s", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16" if fp16 else "f32", "data": "synthetic",
           "config": {"workload": workload, "global_batch": batch * world, "parallelism": "batch-sharded x%d, RCCL all-gather of detection records" % world,
                      "proposals_per_image": [int(c) for c in gpu["proposal_count"]], "detections_per_image": [int(c) for c in gpu["det.count"]]},
           "roofline": roof,
           "step_ms": {"mean": round(elapsed / steps * 1e3, 3), "p50": round(pct(step_ms, 0.5), 3), "p90": round(pct(step_ms, 0.9), 3),
                       "note": "intervals between consecutive per-step completion events on the results stream (pipelined multi-stream run)" +
                               ("; the RCCL all-gather of the step runs on that stream in front of its mark, so it is inside these intervals" if gather is not None else "")},
           "mean_ms_per_image": round(elapsed / steps * 1e3 / batch, 3), "p50_ms_per_image": round(pct(step_ms, 0.5) / batch, 3)}
    out["value_note"] = "SURVEY 8(d): the H2D of every batch is inside the step (%.1f MB of uint8 images through pinned memory on a copy stream, double-buffered; mean subtraction + to_image_list padding on the engine's stream); results stay on the device (value_e2e ships them)" % (flat.nbytes / 1e6)
    if resident_elapsed is not None:
        out["value_resident"] = round(batch * world * steps / resident_elapsed, 3)
        out["value_incl_h2d_f32"] = round(batch * world * steps / h2d_elapsed, 3)
        out["h2d_note"] = "value_resident: the same K steps with the batch already in HBM; value_incl_h2d_f32: with the host-normalised fp32 batch (%.1f MB) uploaded every step" % (x.nbytes / 1e6)
    if e2e is not None:
        out["value_e2e"] = round(batch * world * steps / e2e["elapsed"], 3)
        out["e2e"] = {"ms_per_step": round(e2e["elapsed"] / steps * 1e3, 3), "record_block_bytes": int(e2e["record_bytes"]), "rle_chars_per_step": int(e2e["chars_per_step"]),
                      "blocks_collected": int(e2e["blocks"]), "device_rle_equals_host_encoder": bool(e2e["rle_ok"]), "masks_checked": int(e2e["rle_checked"]),
                      "note": "uint8 upload -> forward -> Masker paste at 800x1333 -> pycocotools RLE on the device -> one fixed-size record block per step in pinned host memory%s while the next step runs; the %d MB of uint8 mask planes never leave the GPU" % (
                          " / all-gathered over RCCL" if gather is not None else "", int(batch * 100 * 800 * 1333 / 1e6))}
    if rccl:
        out.update(rccl)
    traffic = None
    if not c4 and not fp16 and depth == 50 and batch == 2:
        traffic = hbm_traffic("hbm_stage_traffic_maskrcnn.json")
    elif not c4 and fp16 and depth == 101 and batch == 8:
        traffic = hbm_traffic("hbm_stage_traffic_r101f16.json")
    hbm = hbm_rooflines(ops, steps, traffic)
    if rle_op:
        hbm += hbm_rooflines(rle_op, 5, traffic)   # (the front end's counter bytes are per launch, as its algorithmic bytes are)
    out["roofline_hbm"] = hbm
    out["roofline_hbm_note"] = ("SURVEY 8(d) HBM-bound stages, per step of %d images, in the single-stream pass of `roofline`: achieved = ALGORITHMIC bytes / "
                                "HIP-event time on the launching stream; peak 8 TB/s (spec)" % batch)
    if not a.no_latency:
        model.upload(x[:1], hw[:1])
        out["bs1"] = latency_pass(model, lambda: (model.forward_device(1), model.paste_device(800, 1333)), iters=11, drop=3)
        f1, m1, l1, _ = roofline_pass(model, lambda i: (model.forward_device(1), model.paste_device(800, 1333)), full_sync, 10, a.single_stream)
        out["bs1"]["roofline"] = {"bound": "mfma", "achieved": round(f1 / (m1 * 1e-3) / 1e12, 2) if m1 > 0 else 0.0, "peak": peak, "unit": "TFLOP/s",
                                  "frac": round(f1 / (m1 * 1e-3) / 1e12 / peak, 4) if m1 > 0 else 0.0, "conv_ms_per_image": round(m1 / 10, 3),
                                  "pass": "10 single-stream bs=1 steps, HIP events around every conv launch"}
        if not model.fp16:
            model.set_param("conv_split_k", 1.0)
            sk = latency_pass(model, lambda: (model.forward_device(1), model.paste_device(800, 1333)), iters=11, drop=3)
            model.set_param("conv_split_k", 0.0)
            out["bs1"]["conv_split_k"] = dict(sk, note="engine parameter conv_split_k = 1 (default 0): see the Yolact block")
        model.upload(x, hw)
        if not summary:
            model.set_param("timing", 1.0)
            run(); model.sync()
            out["stage_ms_bs%d" % batch] = {k: round(v, 3) for k, v in model.timings()}
            model.set_param("timing", 0.0)
    if world == 1 and not a.no_cpu_baseline and not fp16 and depth == 50 and not c4:
        from oracle.maskrcnn_ref import MaskRCNNRef
        ncpu = min(len(os.sched_getaffinity(0)), 16)
        set_omp_threads(ncpu)
        ref = MaskRCNNRef(sd)
        tc = time.perf_counter()
        rd = ref.forward(x, hw)  # the whole bench batch: timing sample AND parity reference
        pasted = [MaskRCNNRef.paste(r, 800, 1333)[0] for r in rd]
        tcpu = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": round(batch / tcpu, 4), "unit": "img/s", "cores": ncpu, "kind": "port",
                               "sample": "one pass over the %d images of the bench batch, oracle/ C+numpy restatement (AVX2 FMA + OpenMP, %d threads), %.1f s" % (batch, ncpu, tcpu)}
        if not summary:
            set_omp_threads(1)
            tc = time.perf_counter()
            r1 = ref.forward(x[:1], hw[:1]); MaskRCNNRef.paste(r1[0], 800, 1333)
            t1 = time.perf_counter() - tc
            set_omp_threads(ncpu)
            out["cpu_baseline"]["value_1thread"] = round(1.0 / t1, 4)
            out["cpu_baseline"]["sample_1thread"] = "1 image of the bench batch, 1 thread, %.1f s" % t1
        ok = True
        for i in range(batch):
            r = rd[i]
            c = len(r["score"])
            ok &= int(gpu["det.count"][i]) == c and int(gpu["proposal_count"][i]) == len(r["proposals"])
            ok &= np.array_equal(gpu["proposals"][i, : len(r["proposals"])], r["proposals"])
            ok &= all(np.array_equal(gpu[g][i, :c], r[o]) for g, o in (("det.score", "score"), ("det.label", "label"), ("det.box", "box"), ("det.mask28", "mask28")))
            ok &= np.array_equal(gpu["det.masks"][i, :c], pasted[i])
        out["parity_vs_oracle_on_bench_batch"] = bool(ok)
        out["parity_note"] = "%d images: proposals, detection count, score, label, box, 28x28 masks and the masks pasted at 800x1333 all bit-equal" % batch
    out["stream_layout"] = stream_layout(model)
    if summary:
        keep = {"workload": out["config"]["workload"], "img_per_s": out["value"], "batch": batch, "ms_per_step": out["ms_per_step"], "steps": steps, "warmup": warmup,
                "dtype": out["dtype"], "step_ms": out["step_ms"], "bs1": out.get("bs1"), "roofline_hbm": out.get("roofline_hbm"),
                "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "conv_ms_per_step", "launches_per_step", "algorithmic_gflop_per_step", "traffic", "traffic_note")},
                "detections_per_image": out["config"]["detections_per_image"], "proposals_per_image": out["config"]["proposals_per_image"],
                "stream_layout": out["stream_layout"]}
        if summary == "r50":
            keep.update({"value_resident": out.get("value_resident"), "value_e2e": out.get("value_e2e"), "e2e": out.get("e2e"),
                         "cpu_baseline": out.get("cpu_baseline"), "parity_vs_oracle_on_bench_batch": out.get("parity_vs_oracle_on_bench_batch"),
                         "north_star_target": "Mask R-CNN R50-FPN 1333x800 bs=1 >= 30 img/s: bs1.img_per_s"})
        else:
            keep["parity_note"] = "fp16 is tolerance-parity, not index-exact: tests/test_maskrcnn_e2e_gpu.py::test_maskrcnn_r101_fp16_bs8_full_size (-m gpu) checks this shape against the fp16-emulating oracle"
        return keep, model, gather
    return out, model, gather


def main():
    global _JSON_FD
    a = parse()
    global ALLOW_STALE_TRAFFIC
    ALLOW_STALE_TRAFFIC = bool(a.allow_stale_traffic)
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    dist = Dist(a)
    from isegmi import _ffi
    if _ffi.device_count() < 1:
        raise SystemExit("no HIP device visible: bench.py measures the HIP path only (no CPU fallback)")
    if a.f16_mfma_shape >= 0:
        _ffi.set_f16_mfma_shape(a.f16_mfma_shape)
    box = None
    if not a.no_box and dist.rank == 0:
        _ffi.set_device(dist.local_rank)
        box = box_calibration()
    if a.model == "maskrcnn":
        out, net, gather = bench_maskrcnn(a, dist)
    else:
        out, net, gather = bench_yolact(a, dist)
    if out is not None and a.model == "yolact":
        out["stream_layout"] = stream_layout(net)
    net.close()
    if out is not None and a.model == "yolact" and dist.world == 1 and not dist.on and not a.no_maskrcnn and a.yolact_config == "resnet50" and not a.fp16:
        m, model, _ = bench_maskrcnn(a, dist, summary="r50")
        out["maskrcnn"] = m
        model.close()
        if not a.no_r101f16:
            m, model, _ = bench_maskrcnn(a, dist, summary="r101f16")
            out["maskrcnn_r101_fp16"] = m
            model.close()
    if out is not None:
        if box is not None:
            out["box"] = box
        if dist.info():
            out.update(dist.info())
        emit(json.dumps(out))
    dist.close()


if __name__ == "__main__":
    main()
