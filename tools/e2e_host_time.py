"""Where the host time of one e2e step goes (upload -> forward -> masks -> RLE -> record block -> pinned memory -> unpack), and the device step
interval next to the plain pipelined step's: python tools/e2e_host_time.py [yolact|maskrcnn] [fp16] [d101] [bsN] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
from isegmi.pipeline import RecordPipeline

model = sys.argv[1] if len(sys.argv) > 1 else "yolact"
fp16 = "fp16" in sys.argv
steps = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 60
rng = np.random.default_rng(0)
if model == "yolact":
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact
    bs = 8
    net = Yolact(yolact_state_dict(1234), max_batch=bs, fp16=fp16)
    raw = rng.integers(0, 256, (bs, 550, 550, 3), dtype=np.uint8)
    pin = _ffi.PinnedBuffer(raw.shape, np.uint8); pin.array[...] = raw
    up = lambda slot: net.upload_u8_async(pin, bs, 550, 550, slot)
    fwd = lambda slot: net.forward_device(bs, slot)
    post = lambda: net.postprocess_device(550, 550)
else:
    from isegmi.weights import maskrcnn_state_dict
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig
    bs = ([int(a[2:]) for a in sys.argv if a.startswith("bs")] or [2])[0]
    depth = 101 if "d101" in sys.argv else 50
    net = MaskRCNN(maskrcnn_state_dict(1234, depth), 800, 1344, cfg=MaskRCNNConfig(depth=depth), max_batch=bs, fp16=fp16)
    raw = rng.integers(0, 256, (bs, 800, 1333, 3), dtype=np.uint8)
    pin = _ffi.PinnedBuffer(raw.shape, np.uint8); pin.array[...] = raw
    hw = [(800, 1333)] * bs
    up = lambda slot: net.upload_u8_async(pin, hw, slot)
    fwd = lambda slot: net.forward_device(bs, slot)
    post = lambda: net.paste_device(800, 1333)
for mode in ("plain", "e2e"):
    pipe = RecordPipeline(net, bs) if mode == "e2e" else None
    sub = {"fence": [], "pack": [], "dl_async": [], "dl_wait": [], "unpack": []}
    if pipe:  # the calls inside submit(), timed one by one
        def timed(obj, name, key):
            f = getattr(obj, name)
            def g(*a, **k):
                t = time.perf_counter(); r = f(*a, **k); sub[key].append(time.perf_counter() - t); return r
            setattr(obj, name, g)
        timed(net, "download_fence", "fence"); timed(net, "pack_coco_records", "pack"); timed(net, "download_async", "dl_async")
        timed(net, "download_wait", "dl_wait"); timed(pipe, "_unpack", "unpack")
    up(0)
    for i in range(6):
        up((i + 1) & 1); fwd(i & 1); post()
        if pipe: net.rle_device(); pipe.submit(i)
    if pipe: pipe.flush()
    net.sync(); net.step_times(); net.mark_step()
    T = {k: [] for k in ("upload", "forward", "post", "rle", "submit", "wait_mark")}
    up(0)
    t0 = time.perf_counter()
    for i in range(steps):
        a = time.perf_counter(); up((i + 1) & 1)
        b = time.perf_counter(); fwd(i & 1)
        c = time.perf_counter(); post()
        d = time.perf_counter()
        if pipe: net.rle_device()
        e = time.perf_counter()
        if pipe: pipe.submit(i)
        f = time.perf_counter()
        net.mark_step()
        if not pipe: net.wait_mark(1)
        g = time.perf_counter()
        for k, v in zip(T, (b - a, c - b, d - c, e - d, f - e, g - f)): T[k].append(v)
    if pipe: pipe.flush()
    net.sync()
    el = time.perf_counter() - t0
    sm = net.step_times()
    print("%s %s%s: %.3f ms/step  %.1f img/s" % (model, "fp16 " if fp16 else "", mode, el / steps * 1e3, bs * steps / el))
    for k in T:
        print("   host %-9s mean %6.0f us  p50 %6.0f  max %6.0f" % (k, np.mean(T[k]) * 1e6, np.median(T[k]) * 1e6, np.max(T[k]) * 1e6))
    print("   host total per step %.0f us;  device step intervals p10 %.2f p50 %.2f p90 %.2f max %.2f ms" % (
        sum(np.mean(T[k]) for k in T) * 1e6, *np.percentile(sm, [10, 50, 90, 100])))
    if pipe:
        for k in sub:
            print("   submit/%-9s mean %6.0f us  max %6.0f" % (k, np.mean(sub[k][-steps:]) * 1e6, np.max(sub[k][-steps:]) * 1e6))
        pipe.close()
net.close()
