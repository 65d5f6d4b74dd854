import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
from isegmi.weights import yolact_state_dict
from isegmi.yolact import Yolact, fast_base_transform
_ffi.set_device(0)
rng = np.random.default_rng(1)
net = Yolact(yolact_state_dict(1234), max_batch=8)
net.upload(fast_base_transform(rng.integers(0, 256, (8, 550, 550, 3), dtype=np.uint8)))
def run(): net.forward_device(8); net.postprocess_device(550, 550)
for mode in [("multi_stream", 1.0, "pipeline_heads", 1.0), ("multi_stream", 1.0, "pipeline_heads", 0.0), ("multi_stream", 0.0, "pipeline_heads", 0.0)] * 2:
    net.set_param(mode[0], mode[1]); net.set_param(mode[2], mode[3])
    for _ in range(3): run()
    net.sync(); t0 = time.perf_counter()
    for _ in range(20): run()
    net.sync(); dt = (time.perf_counter() - t0) / 20
    print(mode, "%.3f ms/step %.1f img/s" % (dt * 1e3, 8 / dt), flush=True)
