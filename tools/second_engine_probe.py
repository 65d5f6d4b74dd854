"""Does an engine run slower when another engine was created (and closed) earlier in the same process?  (stream -> hardware-queue placement)
   python tools/second_engine_probe.py [n_dummy_engines]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
from isegmi.weights import maskrcnn_state_dict, yolact_state_dict
from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig
from isegmi.yolact import Yolact

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 0
keep = "keep" in sys.argv
import ctypes as _C
_nf = int(next((a.split("=")[1] for a in sys.argv if a.startswith("foreign=")), "0"))
_ffi.set_device(0)
_hip = _C.CDLL("libamdhip64.so")
_foreign = []
for _i in range(_nf):
    _st = _C.c_void_p(); assert _hip.hipStreamCreate(_C.byref(_st)) == 0; _foreign.append(_st)
dummies = []
for i in range(nd):
    y = Yolact(yolact_state_dict(1234), max_batch=1)
    if keep: dummies.append(y)
    else: y.close()
rng = np.random.default_rng(0)
bs = 2
net = MaskRCNN(maskrcnn_state_dict(1234, 50), 800, 1344, cfg=MaskRCNNConfig(depth=50), max_batch=bs)
raw = rng.integers(0, 256, (bs, 800, 1333, 3), dtype=np.uint8)
pin = _ffi.PinnedBuffer(raw.shape, np.uint8); pin.array[...] = raw
hw = [(800, 1333)] * bs
net.upload_u8_async(pin, hw, 0)
def loop(n):
    for i in range(n):
        net.upload_u8_async(pin, hw, (i + 1) & 1); net.forward_device(bs, i & 1); net.paste_device(800, 1333); net.mark_step(); net.wait_mark(1)
loop(10); net.sync()
t0 = time.perf_counter(); loop(50); net.sync(); el = time.perf_counter() - t0
print("foreign streams %d, dummy engines before: %d (%s)  Mask R-CNN bs=2: %.3f ms/step  %.1f img/s" % (_nf, nd, "kept" if keep else "closed", el / 50 * 1e3, bs * 50 / el))
