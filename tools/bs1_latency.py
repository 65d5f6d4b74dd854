"""Yolact bs=1 latency: p50 with / without the multi-stream engine and the per-stage times (dev tool)."""
import os, sys, time; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi.weights import yolact_state_dict
from isegmi.yolact import Yolact, fast_base_transform
net = Yolact(yolact_state_dict(1234), max_batch=1)
x = fast_base_transform(np.random.default_rng(1).uniform(0,255,(1,550,550,3)).astype(np.float32))
net.upload(x)
for ms in (1.0, 0.0):
    net.set_param("multi_stream", ms)
    for _ in range(3): net.forward_device(1); net.postprocess_device(550,550)
    net.sync()
    lat=[]
    for _ in range(15):
        net.sync(); t=time.perf_counter(); net.forward_device(1); net.postprocess_device(550,550); net.sync(); lat.append((time.perf_counter()-t)*1e3)
    print("multi_stream", ms, "p50 %.3f ms"%sorted(lat)[7])
net.set_param("multi_stream", 0.0); net.set_param("timing", 1.0)
net.forward_device(1); net.postprocess_device(550,550); net.sync()
print({k: round(v,3) for k,v in net.timings()})
