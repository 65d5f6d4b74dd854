"""Dev tool: time the fused fp16 stem (conv 7x7/2 + BN + ReLU + max-pool) against the two launches' conv, with the timing-only flag bits
(4 no pooling, 8 no epilogue, 16 no MFMAs / epilogue, 32 no input loads).  argv: N H W [flags,...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
N, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 800, 1344)
FLAGS = [int(f) for f in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 4, 8, 12, 16, 32, 60]
rng = np.random.default_rng(0)
x = rng.uniform(-120, 130, (N, H, W, 3)).astype(np.float32)
w = (rng.standard_normal((64, 7, 7, 4)) * 0.1).astype(np.float32); w[..., 3] = 0
d = _ffi.make_conv_desc(N, H, W, 4, 64, 7, 7, 2, 3, 1, 0)
dx = _ffi.DeviceBuffer.from_numpy(x); dh = _ffi.DeviceBuffer((N, H + 6, (W + 7) & ~1, 4), np.float16)
_ffi.check(_ffi.lib().isegmi_op_pad_c3_to_f16_halo(dx.ptr, N, H, W, dh.ptr, None))
dw = _ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights_f16(d, w))
ds = _ffi.DeviceBuffer.from_numpy(np.full(64, 0.01, np.float32)); dsh = _ffi.DeviceBuffer.from_numpy(np.zeros(64, np.float32))
hc, wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
dc = _ffi.DeviceBuffer((N, hc, wc, 64), np.float16); do = _ffi.DeviceBuffer((N, (hc - 1) // 2 + 1, (wc - 1) // 2 + 1, 64), np.float16)
def timeit(run, n=30):
    for _ in range(3): run()
    _ffi.sync(); t0 = time.perf_counter()
    for _ in range(n): run()
    _ffi.sync(); return (time.perf_counter() - t0) / n * 1e3
print("conv launch alone: %.3f ms" % timeit(lambda: _ffi.check(_ffi.lib().isegmi_op_conv2d_f16(C.byref(d), dh.ptr, dw.ptr, ds.ptr, dsh.ptr, None, dc.ptr, 0, None))))
for f in FLAGS:
    print("fused flags %2d: %.3f ms" % (f, timeit(lambda: _ffi.check(_ffi.lib().isegmi_op_stem_pool_f16(N, H, W, dh.ptr, dw.ptr, ds.ptr, dsh.ptr, do.ptr, f, None)))), flush=True)
