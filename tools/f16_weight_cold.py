"""Dev tool: fp16 conv launches at res4 / res5 shapes of R101 bs 8 with L2-warm weights (one buffer, back to back) against L2-cold ones (rotating over
> 64 MB of weight buffers: inside the model every layer's weights are cold in L2 and come from the Infinity Cache / HBM); activations rotate in both."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
rng = np.random.default_rng(0)
SH = [(8, 50, 84, 1024, 256, 1, 0, 0), (8, 50, 84, 256, 256, 3, 1, 0), (8, 50, 84, 256, 1024, 1, 0, 1), (8, 25, 42, 2048, 512, 1, 0, 0), (8, 25, 42, 512, 512, 3, 1, 0), (8, 25, 42, 512, 2048, 1, 0, 1),
      (8, 100, 168, 256, 256, 3, 1, 0), (8, 200, 336, 256, 256, 3, 1, 0)]
for (N, H, W, Cin, Cout, R, pad, res) in SH:
    wbytes = Cout * R * R * Cin * 2
    nw = max(2, int(80e6 // wbytes) + 1)
    d = _ffi.make_conv_desc(N, H, W, Cin, Cout, R, R, 1, pad, 1, 0)
    w = (rng.standard_normal((Cout, R, R, Cin)) * 0.05).astype(np.float32)
    pk = _ffi.pack_conv_weights_f16(d, w)
    ws = [_ffi.DeviceBuffer.from_numpy(pk) for _ in range(min(nw, 200))]
    xs = [_ffi.DeviceBuffer.from_numpy(np.maximum(rng.standard_normal((N, H, W, Cin)), 0).astype(np.float16)) for _ in range(3)]
    os_ = [_ffi.DeviceBuffer((N, H, W, Cout), np.float16) for _ in range(3)]
    rs = [_ffi.DeviceBuffer.from_numpy(rng.standard_normal((N, H, W, Cout)).astype(np.float16)) for _ in range(3)] if res else None
    def run(i, cold):
        _ffi.check(_ffi.lib().isegmi_op_conv2d_f16(C.byref(d), xs[i % 3].ptr, ws[i % len(ws) if cold else 0].ptr, None, None, rs[i % 3].ptr if res else None, os_[i % 3].ptr, 0, None))
    out = []
    for cold in (False, True):
        n = 300
        for i in range(len(ws)): run(i, cold)
        _ffi.sync(); t0 = time.perf_counter()
        for i in range(n): run(i, cold)
        _ffi.sync(); out.append((time.perf_counter() - t0) / n * 1e6)
    print("M=%-6d K=%-5d Cout=%-5d res %d  w %.2f MB x %d: warm %.1f us  cold %.1f us" % (N * H * W, R * R * Cin, Cout, res, wbytes / 1e6, len(ws), out[0], out[1]), flush=True)
