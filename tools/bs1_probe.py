"""Yolact / Mask R-CNN bs=1 latency probe (dev tool): p50 of forward + postprocess, multi-stream and single-stream, stage marks."""
import ctypes as C, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
from isegmi.weights import yolact_state_dict
from isegmi.yolact import Yolact, fast_base_transform
rng = np.random.default_rng(1)
net = Yolact(yolact_state_dict(1234), max_batch=1)
for kv in sys.argv[1:]:
    net.set_param(kv.split("=")[0], float(kv.split("=")[1]))
net.upload(fast_base_transform(rng.uniform(0, 255, (1, 550, 550, 3)).astype(np.float32)))
def p50(n=25):
    lat = []
    for i in range(n):
        net.sync(); t = time.perf_counter(); net.forward_device(1); net.postprocess_device(550, 550); net.sync(); lat.append((time.perf_counter() - t) * 1e3)
    lat = sorted(lat[5:]); return lat[len(lat) // 2]
print("multi-stream p50 %.3f ms" % p50())
net.set_param("timing", 1.0); net.forward_device(1); net.postprocess_device(550, 550); net.sync()
print("stage marks:", [(k, round(v, 3)) for k, v in net.timings()]); net.set_param("timing", 0.0)
net.set_param("multi_stream", 0.0)
print("single-stream p50 %.3f ms" % p50())
net.set_param("graph", 1.0); net.set_param("multi_stream", 1.0)
print("graph p50 %.3f ms" % p50())
