#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RoI/mask inference hot path.

`python bench.py --gpus N --steps K --warmup W`  (N>1: launched by torch.distributed.run, one rank per GPU).

A "step" = one pass of the hot path over one batch of synthetic input that is already resident
in HBM: Yolact R50-FPN 550x550, bs=8 per GPU (BASELINE.json configs[1]): backbone -> FPN ->
protonet + prediction heads -> Detect (softmax, decode, fast-NMS) -> postprocess (mask assembly
at 550x550, uint8) [-> RCCL all-gather of detection records when N>1].  `--model maskrcnn`
switches to Mask R-CNN R50-FPN 1333x800 bs=2 (configs[2]) once that path is built.

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "instancesegmentation-jittor_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak (spec)
YOLACT_GFLOP_PER_IMAGE = 118.28  # algorithmic conv work per 550x550 image


def pmc_traffic(name):
    """HBM bytes per conv launch from the committed PMC summary (tools/pmc_summary.py), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)["conv_mfma_kernel_all"]["hbm_bytes_per_launch"]
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
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
    ap.add_argument("--single-stream", action="store_true", help="profiling aid: run the timed region on one stream too, so that "
                    "rocprofv3 per-kernel durations are not inflated by overlapping launches (throughput drops ~20 %%)")
    return ap.parse_args()


_JSON_FD = None


def emit(line):
    """The ONE JSON line goes to the process's original stdout; fd 1 itself points at stderr for the whole run, so
    that library banners (RCCL version banner, gloo's connection message ...) cannot precede or follow it."""
    os.write(_JSON_FD if _JSON_FD is not None else 1, (line + "\n").encode())


def main():
    global _JSON_FD
    a = parse()
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    if a.model == "maskrcnn":
        return main_maskrcnn(a)
    a.batch = a.batch or 8
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))

    force_dist = os.environ.get("ISEGMI_BENCH_FORCE_DIST") == "1"  # exercise the N>1 code path on a 1-GPU box
    dist = None
    if world > 1 or force_dist:
        # torch.distributed is plumbing only (rendezvous, barrier, max-reduce of the wall time):
        # CPU/gloo, so torch never touches the GPU.  The data-path collective is RCCL in libisegmi.
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from isegmi import _ffi
    from isegmi.dist import RcclGather, record_bytes
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, fast_base_transform

    if _ffi.device_count() < 1:
        raise SystemExit("no HIP device visible: bench.py measures the HIP path only (no CPU fallback)")
    from isegmi.yolact import YolactConfig
    ycfg = {"resnet50": YolactConfig(), "base": YolactConfig.base(), "im700": YolactConfig.im700(),
            "plus_resnet50": YolactConfig.plus_resnet50(), "plus_base": YolactConfig.plus_base(), "darknet53": YolactConfig.darknet53()}[a.yolact_config]
    yname = {"resnet50": "Yolact R50-FPN", "base": "Yolact R101-FPN (yolact_base)", "im700": "Yolact R101-FPN 700 (yolact_im700)",
             "plus_resnet50": "YOLACT++ R50-FPN (DCNv2, 9 anchors, mask re-scoring)", "plus_base": "YOLACT++ R101-FPN (DCNv2 every 3rd block, 9 anchors, mask re-scoring)",
             "darknet53": "Yolact Darknet53-FPN (yolact_darknet53)"}[a.yolact_config]
    sd = yolact_state_dict(1234, ycfg.depth, ycfg.num_priors, ycfg.dcn_layers, ycfg.dcn_interval, ycfg.use_maskiou, ycfg.backbone)
    net = Yolact(sd, ycfg, max_batch=a.batch, device=local_rank, fp16=a.fp16)
    ypeak = 2500.0 if a.fp16 else PEAK_F32_MFMA_TFLOPS
    if a.single_stream:
        net.set_param("multi_stream", 0.0)
    size = net.size
    rng = np.random.default_rng(20261003 + rank)
    raw = rng.uniform(0, 255, (a.batch, size, size, 3)).astype(np.float32)
    if ycfg.backbone == "darknet53":
        from isegmi.yolact import darknet_base_transform
        imgs = darknet_base_transform(raw)
    else:
        imgs = fast_base_transform(raw)
    net.upload(imgs)

    gather = None
    if world > 1 or force_dist:
        import torch
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(RcclGather.unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(uid, 0)
        gather = RcclGather(rank, world, bytes(uid.numpy().tobytes()), record_bytes(a.batch))

    def step():
        net.forward_device(a.batch)
        net.postprocess_device(size, size)
        if gather is not None:
            gather.gather_from(net)

    def full_sync():
        net.sync()
        if gather is not None:
            gather.wait()
        _ffi.sync()

    for _ in range(max(a.warmup, 1 if gather is not None else 0)):
        step()
    full_sync()
    import ctypes as C
    f, m, l = C.c_double(), C.c_double(), C.c_int64()

    if dist is not None:
        dist.barrier()
    full_sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    full_sync()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        import torch
        te = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # roofline pass: the same K steps again, single-stream, every conv launch bracketed by HIP events on the
    # engine stream (per-launch durations are not meaningful while launches from several streams overlap)
    net.set_param("multi_stream", 0.0)
    net.set_param("conv_timing", 1.0)
    _ffi.check(_ffi.lib().isegmi_engine_conv_stats(net._h, C.byref(f), C.byref(m), C.byref(l)))  # reset
    for _ in range(a.steps):
        step()
    full_sync()
    _ffi.check(_ffi.lib().isegmi_engine_conv_stats(net._h, C.byref(f), C.byref(m), C.byref(l)))
    net.set_param("conv_timing", 0.0)
    net.set_param("multi_stream", 0.0 if a.single_stream else 1.0)
    conv_flops, conv_ms, conv_launches = f.value, m.value, l.value

    counts = net.fetch("det.count", a.batch)
    total_images = a.batch * world * a.steps
    value = total_images / elapsed

    out = None
    if rank == 0:
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        out = {
            "metric": "images/sec (%s %dx%d, bs=%d per GPU, %s)" % (yname, size, size, a.batch, "fp16 storage / f16 MFMA, fp32 accumulate (optional mode, not configs[1])" if a.fp16 else "fp32"),
            "value": round(value, 2),
            "unit": "img/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16" if a.fp16 else "f32",
            "data": "synthetic",
            "config": {"workload": "%s %dx%d bs=%d/GPU random weights: backbone+FPN+protonet+heads+Detect(fast-NMS)+%dx%d mask assembly%s" % (
                           yname, size, size, a.batch, size, size, " (BASELINE configs[1])" if a.yolact_config == "resnet50" else " (variant, not configs[1])"),
                       "global_batch": a.batch * world, "parallelism": "batch-sharded x%d, RCCL all-gather of detections" % world,
                       "detections_per_image_rank0": [int(c) for c in counts]},
            "roofline": {
                "bound": "mfma",
                "kernel": ("conv_f16_glds / conv3x3_f16_strip kernels (all %d conv launches of a step, v_mfma_f32_32x32x16_f16)" if a.fp16 else "conv_mfma_kernel + conv_mfma16_kernel (all %d conv launches of a step; v_mfma_f32_32x32x2_f32 on 64x64 tiles, v_mfma_f32_16x16x4_f32 on the 32x32 blocks of small grids)") % (conv_launches // max(a.steps, 1)),
                "pass": "K single-stream steps right after the timed region, HIP events around every conv launch",
                "achieved": round(achieved, 2),
                "peak": ypeak,
                "unit": "TFLOP/s",
                "frac": round(achieved / ypeak, 4),
                "traffic": None if (a.fp16 or a.yolact_config != "resnet50") else pmc_traffic("r01_pmc_yolact.json"),  # null: no committed PMC pass for this variant
                "traffic_note": "HBM bytes per conv launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (profiles/r01_pmc_yolact.json; FETCH x2 gfx950 correction); not collected live",
                "algorithmic_gflop_per_step": round(conv_flops / max(a.steps, 1) / 1e9, 2),
                "conv_ms_per_step": round(conv_ms / max(a.steps, 1), 3),
                "launches_per_step": conv_launches // max(a.steps, 1),
                "avg_launch_us": round(conv_ms * 1e3 / max(conv_launches, 1), 2),
            },
            "p50_ms_per_image": round(elapsed / a.steps * 1e3 / a.batch, 3),
        }

    # ---- extra: bs=1 latency pass (metric also asks for bs=1 img/s and p50 per-image latency)
    if rank == 0 and not a.no_latency:
        lat = []
        for i in range(13):
            net.sync()
            ts = time.perf_counter()
            net.forward_device(1)
            net.postprocess_device(size, size)
            net.sync()
            lat.append((time.perf_counter() - ts) * 1e3)
        lat = sorted(lat[3:])
        out["bs1"] = {"p50_ms_per_image": round(lat[len(lat) // 2], 3), "img_per_s": round(1e3 / lat[len(lat) // 2], 2)}
        # same pass with the forward replayed as one hipGraph (latency mode: removes the per-launch gaps)
        net.set_param("graph", 1.0)
        lat = []
        for i in range(14):
            net.sync()
            ts = time.perf_counter()
            net.forward_device(1)
            net.postprocess_device(size, size)
            net.sync()
            lat.append((time.perf_counter() - ts) * 1e3)
        lat = sorted(lat[4:])
        out["bs1_hipgraph"] = {"p50_ms_per_image": round(lat[len(lat) // 2], 3), "img_per_s": round(1e3 / lat[len(lat) // 2], 2)}
        net.set_param("graph", 0.0)
        net.set_param("timing", 1.0)
        net.forward_device(a.batch); net.postprocess_device(size, size); net.sync()
        out["stage_ms_bs%d" % a.batch] = {k: round(v, 3) for k, v in net.timings()}
        net.set_param("timing", 0.0)

    # ---- CPU baseline: the oracle restatement on the host cores (rank 0, N=1 only)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle.yolact_ref import YolactRef
        ncpu = min(len(os.sched_getaffinity(0)), 16)  # the GPU box's CPU share for one GPU
        os.environ["OMP_NUM_THREADS"] = str(ncpu)
        ref = YolactRef(sd, max_size=ycfg.max_size, scales=ycfg.pred_scales, depth=ycfg.depth, scales_per_level=ycfg.scales_per_level,
                        square=ycfg.use_square_anchors)
        k = max(1, min(a.cpu_sample, a.batch))
        tc = time.perf_counter()
        done = 0
        dets = None
        while True:  # bounded sample: whole passes over k images until ~10 s of CPU work (cap: 6 passes)
            d_ = ref.forward(imgs[:k])
            for d in d_:
                YolactRef.postprocess(d, size, size)
            dets = dets or d_
            done += k
            tcpu = time.perf_counter() - tc
            if tcpu >= 10.0 or done >= 6 * k:
                break
        out["cpu_baseline"] = {"value": round(done / tcpu, 4), "unit": "img/s", "cores": ncpu, "kind": "port",
                               "sample": "%d images (passes over %d images of the bench batch), oracle/ C+numpy restatement (AVX2 FMA + OpenMP, %d threads), %.1f s" % (done, k, ncpu, tcpu)}
        # the oracle run doubles as a parity check of this very batch
        got = net.fetch("det.prior", a.batch)
        ok = all(np.array_equal(got[i, : len(dets[i]["prior"])], dets[i]["prior"]) for i in range(k))
        if not a.fp16:  # the fp16 mode is tolerance-parity (tests), not index-exact
            out["parity_vs_oracle_on_bench_batch"] = bool(ok)

    if rank == 0:
        emit(json.dumps(out))
    if gather is not None:
        gather.close()
    net.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_maskrcnn(a):
    """Mask R-CNN R50-FPN 1333x800 (padded 800x1344), bs=2 per GPU: BASELINE configs[2]; step = forward + Masker paste."""
    import ctypes as C
    a.batch = a.batch or 2
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from isegmi import _ffi
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig, prepare_images
    from isegmi.weights import maskrcnn_state_dict
    if _ffi.device_count() < 1:
        raise SystemExit("no HIP device visible: bench.py measures the HIP path only (no CPU fallback)")
    if a.c4:
        from isegmi.weights import maskrcnn_c4_state_dict
        assert not a.fp16 and a.depth == 50, "--c4 is R-50, fp32"
        sd, mcfg = maskrcnn_c4_state_dict(1234), MaskRCNNConfig.c4()
    else:
        sd, mcfg = maskrcnn_state_dict(1234, depth=a.depth), MaskRCNNConfig(depth=a.depth)
    rng = np.random.default_rng(20261003 + rank)
    imgs = [rng.uniform(0, 255, (800, 1333, 3)).astype(np.float32) for _ in range(a.batch)]
    x, hw = prepare_images(imgs, mcfg.SIZE_DIVISIBILITY)
    model = MaskRCNN(sd, x.shape[1], x.shape[2], cfg=mcfg, max_batch=a.batch, device=local_rank, fp16=a.fp16)
    if a.single_stream:
        model.set_param("multi_stream", 0.0)
    tag = "R50-C4" if a.c4 else "R%d-FPN" % a.depth
    prec = "fp16 storage / f16 MFMA, fp32 accumulate" if a.fp16 else "fp32"
    peak = 2500.0 if a.fp16 else PEAK_F32_MFMA_TFLOPS  # dense f16 MFMA peak (MI355X_MICROARCH.md) vs f32 MFMA peak
    model.upload(x, hw)
    gather = None
    if world > 1:
        import torch
        from isegmi.dist import RcclGather, maskrcnn_record_bytes
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(RcclGather.unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(uid, 0)
        gather = RcclGather(rank, world, bytes(uid.numpy().tobytes()), maskrcnn_record_bytes(a.batch, M=14 if a.c4 else 28))

    def step():
        model.forward_device(a.batch)
        model.paste_device(800, 1333)
        if gather is not None:
            gather.gather_from(model)

    for _ in range(max(a.warmup, 1 if gather is not None else 0)):
        step()
    model.sync(); _ffi.sync()
    if gather is not None:
        gather.wait()
    f, m, l = C.c_double(), C.c_double(), C.c_int64()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    model.sync()
    if gather is not None:
        gather.wait()
    _ffi.sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        te = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    model.set_param("multi_stream", 0.0)
    model.set_param("conv_timing", 1.0)
    _ffi.check(_ffi.lib().isegmi_engine_conv_stats(model._h, C.byref(f), C.byref(m), C.byref(l)))
    for _ in range(a.steps):
        step()
    model.sync(); _ffi.sync()
    _ffi.check(_ffi.lib().isegmi_engine_conv_stats(model._h, C.byref(f), C.byref(m), C.byref(l)))
    model.set_param("conv_timing", 0.0)
    model.set_param("multi_stream", 0.0 if a.single_stream else 1.0)
    if rank == 0:
        achieved = f.value / (m.value * 1e-3) / 1e12 if m.value > 0 else 0.0
        cnt = model.fetch("det.count", a.batch); pc = model.fetch("proposal_count", a.batch)
        out = {"metric": "images/sec (Mask R-CNN %s 1333x800, bs=%d per GPU, %s)" % (tag, a.batch, prec),
               "value": round(a.batch * world * a.steps / elapsed, 3), "unit": "img/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f16" if a.fp16 else "f32", "data": "synthetic",
               "config": {"workload": ("Mask R-CNN %s 1333x800 (padded 800x1344) bs=%d/GPU random weights, %s: conv1-4 + single-map RPN (6000 -> 1000) + RoIAlign + conv5 head + NMS + shared-extractor mask branch + paste (the README.md:263-273 config; not a BASELINE config)" % (tag, a.batch, prec)) if a.c4 else "Mask R-CNN %s 1333x800 (padded 800x1344) bs=%d/GPU random weights, %s: backbone+FPN+RPN+RoIAlign+box head+NMS+mask head+paste (BASELINE configs[%d])" % (tag, a.batch, prec, 4 if a.fp16 else 2),
                          "global_batch": a.batch * world, "parallelism": "batch-sharded x%d" % world,
                          "proposals_per_image": [int(c) for c in pc], "detections_per_image": [int(c) for c in cnt]},
               "roofline": {"bound": "mfma", "kernel": "conv_mfma_kernel + conv_mfma16_kernel (all conv launches of a step)",
                            "pass": "K single-stream steps right after the timed region, HIP events around every conv launch", "achieved": round(achieved, 2),
                            "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": pmc_traffic("r01_pmc_r101f16.json") if (a.fp16 and a.depth == 101 and a.batch == 8 and not a.c4) else None if (a.fp16 or a.c4 or a.depth != 50 or a.batch != 2) else pmc_traffic("r01_pmc_maskrcnn.json"),
                            "traffic_note": "HBM bytes per conv launch from separate rocprofv3 --pmc passes of this command (tools/profile_round.sh -> profiles/r01_pmc_maskrcnn.json / r01_pmc_r101f16.json; FETCH x2 gfx950 correction); not collected live; null for configurations without a committed PMC pass",
                            "algorithmic_gflop_per_step": round(f.value / a.steps / 1e9, 2), "conv_ms_per_step": round(m.value / a.steps, 3),
                            "launches_per_step": l.value // a.steps, "avg_launch_us": round(m.value * 1e3 / max(l.value, 1), 2)},
               "p50_ms_per_image": round(elapsed / a.steps * 1e3 / a.batch, 3)}
        if not a.no_latency:
            lat = []
            model.upload(x[:1], hw[:1])
            for i in range(9):
                model.sync(); ts = time.perf_counter()
                model.forward_device(1); model.paste_device(800, 1333); model.sync()
                lat.append((time.perf_counter() - ts) * 1e3)
            lat = sorted(lat[2:])
            out["bs1"] = {"p50_ms_per_image": round(lat[len(lat) // 2], 3), "img_per_s": round(1e3 / lat[len(lat) // 2], 2)}
            model.set_param("graph", 1.0)  # the forward replayed as one hipGraph (latency mode)
            lat = []
            for i in range(11):
                model.sync(); ts = time.perf_counter()
                model.forward_device(1); model.paste_device(800, 1333); model.sync()
                lat.append((time.perf_counter() - ts) * 1e3)
            lat = sorted(lat[4:])
            out["bs1_hipgraph"] = {"p50_ms_per_image": round(lat[len(lat) // 2], 3), "img_per_s": round(1e3 / lat[len(lat) // 2], 2)}
            model.set_param("graph", 0.0)
            model.upload(x, hw)
            model.set_param("timing", 1.0)
            step(); model.sync()
            out["stage_ms_bs%d" % a.batch] = {k: round(v, 3) for k, v in model.timings()}
            model.set_param("timing", 0.0)
        if world == 1 and not a.no_cpu_baseline and not a.fp16 and a.depth == 50 and not a.c4:
            from oracle.maskrcnn_ref import MaskRCNNRef
            ncpu = min(len(os.sched_getaffinity(0)), 16)
            os.environ["OMP_NUM_THREADS"] = str(ncpu)
            ref = MaskRCNNRef(sd)
            tc = time.perf_counter()
            done = 0
            d = None
            while True:  # bounded sample: whole images until ~10 s of CPU work (cap 6)
                d_ = ref.forward(x[:1], hw[:1])
                MaskRCNNRef.paste(d_[0], 800, 1333)
                d = d or d_
                done += 1
                tcpu = time.perf_counter() - tc
                if tcpu >= 10.0 or done >= 6:
                    break
            out["cpu_baseline"] = {"value": round(done / tcpu, 4), "unit": "img/s", "cores": ncpu, "kind": "port",
                                   "sample": "%d passes over 1 image of the bench batch, oracle/ C+numpy restatement (AVX2 FMA + OpenMP, %d threads), %.1f s" % (done, ncpu, tcpu)}
            got = model.fetch("det.box", 1)[0]
            out["parity_vs_oracle_on_bench_batch"] = bool(np.array_equal(got[: len(d[0]["box"])], d[0]["box"]))
        emit(json.dumps(out))
    model.close()
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
