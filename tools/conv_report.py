"""Per-layer conv report on the GPU (dev tool): python tools_conv_report.py [batch] [tile]"""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
from isegmi.weights import yolact_state_dict
from isegmi.yolact import Yolact, fast_base_transform
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = Yolact(yolact_state_dict(1234), max_batch=bs)
net.set_param("multi_stream", 0.0)
if len(sys.argv) > 2: net.set_param("conv_tile", float(sys.argv[2]))
rng = np.random.default_rng(1)
net.upload(fast_base_transform(rng.uniform(0, 255, (bs, 550, 550, 3)).astype(np.float32)))
for _ in range(2): net.forward_device(bs)
net.sync(); net.set_param("conv_timing", 1.0)
f, m, l = C.c_double(), C.c_double(), C.c_int64()
_ffi.lib().isegmi_engine_conv_stats(net._h, C.byref(f), C.byref(m), C.byref(l))
buf = C.create_string_buffer(1 << 16); _ffi.lib().isegmi_engine_conv_report(net._h, buf, 1 << 16)
R = 5
for _ in range(R): net.forward_device(bs)
net.sync()
_ffi.check(_ffi.lib().isegmi_engine_conv_report(net._h, buf, 1 << 16))
rows = [r.split("\t") for r in buf.value.decode().strip().split("\n")]
rows.sort(key=lambda r: -float(r[2]))
tot = sum(float(r[2]) for r in rows) / R
print("total conv ms/step %.3f" % tot)
for r in rows:
    print("%-95s %8.2f GF %8.3f ms %7.2f TF/s" % (r[0], float(r[1]) / R, float(r[2]) / R, float(r[3])))
