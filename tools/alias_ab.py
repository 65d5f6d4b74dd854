"""Same-process A/B of the liveness-aliased ResNet activation buffers (engine param alias_buffers): single-stream conv time per step and
pipelined throughput, R101 fp16 bs=8 / R50 fp32 bs=2 / Yolact bs=8.  python tools/alias_ab.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
from isegmi.weights import maskrcnn_state_dict, yolact_state_dict
from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig, prepare_images
from isegmi.yolact import Yolact, fast_base_transform
rng = np.random.default_rng(1)


def measure(net, step, tag):
    for alias in (1.0, 0.0, 1.0, 0.0):
        net.set_param("alias_buffers", alias)
        for _ in range(3): step()
        net.sync()
        t0 = time.perf_counter()
        for _ in range(30): step()
        net.sync()
        thr = (time.perf_counter() - t0) / 30 * 1e3
        net.set_param("multi_stream", 0.0); net.set_param("conv_timing", 1.0)
        f, m, l = C.c_double(), C.c_double(), C.c_int64()
        _ffi.lib().isegmi_engine_conv_stats(net._h, C.byref(f), C.byref(m), C.byref(l))
        for _ in range(5): step()
        net.sync()
        _ffi.lib().isegmi_engine_conv_stats(net._h, C.byref(f), C.byref(m), C.byref(l))
        net.set_param("conv_timing", 0.0); net.set_param("multi_stream", 1.0)
        wb, bb = net.memory()
        print("%-14s alias %d: pipelined %.3f ms/step, single-stream conv %.3f ms/step, buffers %.2f GB" % (tag, alias, thr, m.value / 5, bb / 1e9), flush=True)


x, hw = prepare_images([rng.uniform(0, 255, (800, 1333, 3)).astype(np.float32) for _ in range(8)])
net = MaskRCNN(maskrcnn_state_dict(1234, 101), 800, 1344, cfg=MaskRCNNConfig(depth=101), max_batch=8, fp16=True)
net.upload(x, hw)
measure(net, lambda: (net.forward_device(8), net.paste_device(800, 1333)), "R101 fp16 bs8")
net.close()
net = MaskRCNN(maskrcnn_state_dict(1234, 50), 800, 1344, max_batch=2)
net.upload(x[:2], hw[:2])
measure(net, lambda: (net.forward_device(2), net.paste_device(800, 1333)), "R50 fp32 bs2")
net.close()
net = Yolact(yolact_state_dict(1234), max_batch=8)
net.upload(fast_base_transform(rng.uniform(0, 255, (8, 550, 550, 3)).astype(np.float32)))
measure(net, lambda: (net.forward_device(8), net.postprocess_device(550, 550)), "Yolact bs8")
net.close()
