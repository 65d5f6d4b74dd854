"""Dev tool: bs=1 conv launches with L2-warm weights (one weight buffer, launched back to back) against L2-cold ones (rotating over enough
weight buffers to exceed the 32 MB of L2: inside a model every layer's weights are cold in L2 and come from the Infinity Cache / HBM)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
rng = np.random.default_rng(0)
SH = [(1, 138, 138, 64, 64, 3, 1), (1, 69, 69, 128, 128, 3, 1), (1, 35, 35, 256, 256, 3, 1), (1, 35, 35, 1024, 256, 1, 0), (1, 35, 35, 256, 1024, 1, 0),
      (1, 18, 18, 512, 512, 3, 1), (1, 18, 18, 2048, 512, 1, 0), (1, 69, 69, 256, 256, 3, 1), (1, 138, 138, 256, 256, 3, 1)]
for (N, H, W, Cin, Cout, R, pad) in SH:
    wbytes = Cout * R * R * Cin * 4
    nw = max(2, int(80e6 // wbytes) + 1)
    d = _ffi.make_conv_desc(N, H, W, Cin, Cout, R, R, 1, pad, 1, 0)
    w = (rng.standard_normal((Cout, R, R, Cin)) * 0.05).astype(np.float32)
    pk = _ffi.pack_conv_weights(d, w)
    ws = [_ffi.DeviceBuffer.from_numpy(pk) for _ in range(min(nw, 400))]
    x = _ffi.DeviceBuffer.from_numpy(rng.standard_normal((N, H, W, Cin)).astype(np.float32)); o = _ffi.DeviceBuffer((N, H, W, Cout))
    def run(i): _ffi.check(_ffi.lib().isegmi_op_conv2d(C.byref(d), x.ptr, ws[i % len(ws)].ptr, None, None, None, o.ptr, None))
    res = []
    for mode in ("warm", "cold"):
        n = 200
        for i in range(len(ws) if mode == "cold" else 3): run(i if mode == "cold" else 0)
        _ffi.sync(); t0 = time.perf_counter()
        for i in range(n): run(i if mode == "cold" else 0)
        _ffi.sync(); res.append((time.perf_counter() - t0) / n * 1e6)
    print("M=%-6d K=%-5d Cout=%-5d w %.2f MB x %d: warm %.1f us  cold %.1f us" % (N * H * W, R * R * Cin, Cout, wbytes / 1e6, len(ws), res[0], res[1]), flush=True)
