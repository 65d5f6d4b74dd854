"""Experiment (dev tool): does a mid-size conv whose 64x64 tiles fill the chip unevenly (528 tiles on 256 CUs) run faster as TWO concurrent
launches -- the whole-CU-multiple part on the 64x64 v2 kernel, the remainder on the 32x32 16x16x4 kernel -- than as one launch of either?
The two launches write disjoint images of one output tensor; separate streams stand in for a single heterogeneous kernel."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
hip = C.CDLL("libamdhip64.so")
def stream():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; return s
s1, s2 = stream(), stream()
rng = np.random.default_rng(0)
# (images of 16x16 px: 4 m-tiles of 64 rows each)  name, N_total, N_main, Cin, Cout, R
CASES = [("528 tiles K=1024 Cout=256 1x1", 33, 32, 1024, 256, 1), ("528 tiles K=2304 Cout=256 3x3", 33, 32, 256, 256, 3),
         ("616 tiles K=2304 Cout=256 3x3", 39, 32, 256, 256, 3), ("1056 tiles K=512 Cout=128 1x1", 132, 128, 512, 128, 1),
         ("1192 tiles K=512 Cout=128 1x1", 149, 128, 512, 128, 1), ("1192 tiles K=1152 Cout=128 3x3", 149, 128, 128, 128, 3)]
for name, N, Nm, Cin, Cout, R in CASES:
    H = W = 16; pad = R // 2
    x = _ffi.DeviceBuffer.from_numpy(rng.standard_normal((N, H, W, Cin)).astype(np.float32))
    w = (rng.standard_normal((Cout, R, R, Cin)) * 0.05).astype(np.float32)
    out = _ffi.DeviceBuffer((N, H, W, Cout))
    def desc(n, tile): return _ffi.make_conv_desc(n, H, W, Cin, Cout, R, R, 1, pad, 1, tile)
    dw = _ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights(desc(N, 0), w))
    L = _ffi.lib()
    def one(tile):
        d = desc(N, tile)
        return lambda: _ffi.check(L.isegmi_op_conv2d(C.byref(d), x.ptr, dw.ptr, None, None, None, out.ptr, s1))
    dm, dt = desc(Nm, 12 if R == 3 and Cin * 9 >= 2304 else 10), desc(N - Nm, 4)
    xo, oo = Nm * H * W * Cin * 4, Nm * H * W * Cout * 4
    def hybrid():
        _ffi.check(L.isegmi_op_conv2d(C.byref(dm), x.ptr, dw.ptr, None, None, None, out.ptr, s1))
        _ffi.check(L.isegmi_op_conv2d(C.byref(dt), C.c_void_p(x.ptr.value + xo), dw.ptr, None, None, None, C.c_void_p(out.ptr.value + oo), s2))
    res = []
    for label, fn in (("v2", one(10)), ("t6", one(6)), ("t4", one(4)), ("v2 + t4 tail, two streams", hybrid), ("hybrid kernel (tile 13)", one(13)), ("(tile 14)", one(14))):
        for _ in range(5): fn()
        hip.hipDeviceSynchronize(); t0 = time.perf_counter()
        for _ in range(50): fn()
        hip.hipDeviceSynchronize(); res.append("%s %.1f us" % (label, (time.perf_counter() - t0) / 50 * 1e6))
    print(name + ":  " + "   ".join(res), flush=True)
