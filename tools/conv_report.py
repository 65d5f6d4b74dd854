"""Per-layer conv report on the GPU, single stream (dev tool):
   python tools/conv_report.py [batch] [tile] [yolact|maskrcnn] [fp16|fp32] [depth] [H W]     (H W: Mask R-CNN image size before padding, default 800 1333)"""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
model = sys.argv[3] if len(sys.argv) > 3 else "yolact"
fp16 = len(sys.argv) > 4 and sys.argv[4] == "fp16"
depth = int(sys.argv[5]) if len(sys.argv) > 5 else 50
IH, IW = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (800, 1333)
rng = np.random.default_rng(1)
if model == "yolact":
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, fast_base_transform
    net = Yolact(yolact_state_dict(1234), max_batch=bs)
    net.upload(fast_base_transform(rng.uniform(0, 255, (bs, 550, 550, 3)).astype(np.float32)))
    step = lambda: net.forward_device(bs)
else:
    from isegmi.weights import maskrcnn_state_dict
    from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig, prepare_images
    x, hw = prepare_images([rng.uniform(0, 255, (IH, IW, 3)).astype(np.float32) for _ in range(bs)])
    net = MaskRCNN(maskrcnn_state_dict(1234, depth), x.shape[1], x.shape[2], cfg=MaskRCNNConfig(depth=depth), max_batch=bs, fp16=fp16)
    net.upload(x, hw)
    step = lambda: net.forward_device(bs)
net.set_param("multi_stream", 0.0)
if tile: net.set_param("conv_tile", float(tile))
for _ in range(2): step()
net.sync(); net.set_param("conv_timing", 1.0)
f, m, l = C.c_double(), C.c_double(), C.c_int64()
_ffi.lib().isegmi_engine_conv_stats(net._h, C.byref(f), C.byref(m), C.byref(l))
buf = C.create_string_buffer(1 << 18); _ffi.lib().isegmi_engine_conv_report(net._h, buf, 1 << 18)
R = 5
for _ in range(R): step()
net.sync()
_ffi.check(_ffi.lib().isegmi_engine_conv_report(net._h, buf, 1 << 18))
rows = [r.split("\t") for r in buf.value.decode().strip().split("\n")]
rows.sort(key=lambda r: -float(r[2]))
tot = sum(float(r[2]) for r in rows) / R; gf = sum(float(r[1]) for r in rows) / R
print("total conv ms/step %.3f  (%.1f GF, %.1f TF/s)" % (tot, gf, gf / tot))
for r in rows:
    print("%-95s %8.2f GF %8.3f ms %7.2f TF/s" % (r[0], float(r[1]) / R, float(r[2]) / R, float(r[3])))
