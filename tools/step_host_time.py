"""Host-side cost of the calls of one pipelined Yolact step (uint8 upload + forward + postprocess): where a step's host time goes, and
whether any call blocks on the device.  python tools/step_host_time.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
from isegmi.weights import yolact_state_dict
from isegmi.yolact import Yolact

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
net = Yolact(yolact_state_dict(1234), max_batch=8)
rng = np.random.default_rng(0)
raw = rng.integers(0, 256, (8, 550, 550, 3), dtype=np.uint8)
pin = _ffi.PinnedBuffer(raw.shape, np.uint8); pin.array[...] = raw
for mode in ("resident", "u8"):
    net.upload_u8(raw); net.forward_device(8); net.postprocess_device(550, 550); net.sync()
    if mode == "u8":
        net.upload_u8_async(pin, 8, 550, 550, 0)
    T = {"upload": [], "forward": [], "post": []}
    net.sync(); net.step_times(); net.mark_step()
    t0 = time.perf_counter()
    for i in range(steps):
        a = time.perf_counter()
        if mode == "u8":
            net.upload_u8_async(pin, 8, 550, 550, (i + 1) & 1)
        b = time.perf_counter()
        net.forward_device(8, (i & 1) if mode == "u8" else 0)
        c = time.perf_counter()
        net.postprocess_device(550, 550)
        d = time.perf_counter()
        net.mark_step()
        T["upload"].append(b - a); T["forward"].append(c - b); T["post"].append(d - c)
    net.sync()
    el = time.perf_counter() - t0
    sm = net.step_times()
    f = lambda x: "mean %.0f us  p50 %.0f  max %.0f" % (np.mean(x) * 1e6, np.median(x) * 1e6, np.max(x) * 1e6)
    print(mode, "%.3f ms/step  %.1f img/s" % (el / steps * 1e3, 8 * steps / el))
    for k in T:
        print("   host %-8s %s" % (k, f(T[k])))
    print("   device step intervals: p10 %.2f p50 %.2f p90 %.2f max %.2f ms" % tuple(np.percentile(sm, [10, 50, 90, 100])))
    print("   ", " ".join("%.1f" % x for x in sm[:40]))
net.close()
