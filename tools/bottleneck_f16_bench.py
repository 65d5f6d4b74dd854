"""Micro-benchmark (dev tool): fused identity bottleneck (isegmi_op_bottleneck_f16) against the three launches it replaces (auto tiles),
at the res2 / res3 shapes of Mask R-CNN R101 bs=8 and R50 bs=2.  python tools/bottleneck_f16_bench.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
rng = np.random.default_rng(0)
L = _ffi.lib()
REP = int(os.environ.get("REP", "20"))


def timeit(run):
    for _ in range(3): run()
    _ffi.sync(); t0 = time.perf_counter()
    for _ in range(REP): run()
    _ffi.sync(); return (time.perf_counter() - t0) / REP


for (N, H, W, Cin, Cmid) in [(8, 200, 336, 256, 64), (8, 100, 168, 512, 128), (2, 200, 336, 256, 64), (2, 100, 168, 512, 128)]:
    x = np.maximum(rng.standard_normal((N, H, W, Cin)), 0).astype(np.float16)
    dx = _ffi.DeviceBuffer.from_numpy(x)
    ws, descs = [], []
    for (co, r, ci) in ((Cmid, 1, Cin), (Cmid, 3, Cmid), (Cin, 1, Cmid)):
        d = _ffi.make_conv_desc(N, H, W, ci, co, r, r, 1, r // 2, 1, 0)
        w = (rng.standard_normal((co, r, r, ci)) * (2.0 / (r * r * ci)) ** 0.5).astype(np.float32)
        ws += [_ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights_f16(d, w)), _ffi.DeviceBuffer.from_numpy(rng.uniform(0.5, 1.5, co).astype(np.float32)),
               _ffi.DeviceBuffer.from_numpy((rng.standard_normal(co) * 0.1).astype(np.float32))]
        descs.append(d)
    t1 = _ffi.DeviceBuffer((N, H, W, Cmid), np.float16); t2 = _ffi.DeviceBuffer((N, H, W, Cmid), np.float16)
    o3 = _ffi.DeviceBuffer((N, H, W, Cin), np.float16); of = _ffi.DeviceBuffer((N, H, W, Cin), np.float16)
    bd = _ffi.BottleneckDesc(N, H, W, Cin, Cmid, 0)

    def three():
        _ffi.check(L.isegmi_op_conv2d_f16(C.byref(descs[0]), dx.ptr, ws[0].ptr, ws[1].ptr, ws[2].ptr, None, t1.ptr, 0, None))
        _ffi.check(L.isegmi_op_conv2d_f16(C.byref(descs[1]), t1.ptr, ws[3].ptr, ws[4].ptr, ws[5].ptr, None, t2.ptr, 0, None))
        _ffi.check(L.isegmi_op_conv2d_f16(C.byref(descs[2]), t2.ptr, ws[6].ptr, ws[7].ptr, ws[8].ptr, dx.ptr, o3.ptr, 0, None))

    def fused():
        _ffi.check(L.isegmi_op_bottleneck_f16(C.byref(bd), dx.ptr, *[b.ptr for b in ws], of.ptr, None))

    if os.environ.get("VARIANTS"):
        line = "N%d %dx%d C%d:" % (N, H, W, Cin)
        for name, fl_ in (("full", 0), ("no x", 2), ("no res", 4), ("no store", 8), ("no w", 16), ("no x/res", 6), ("no x/res/store", 14), ("no mem", 30), ("no mem, loaders idle", 62), ("no mem, mfma idle", 30 | 64), ("mfma idle (loads real)", 64),
                          ("no mem -c1", 62 | 128), ("no mem -e1", 62 | 256), ("no mem -c2", 62 | 512), ("no mem -e2", 62 | 1024), ("no mem -c3", 62 | 2048), ("no mem -e3", 62 | 4096),
                          ("no mem only barriers+e3", 62 | 128 | 256 | 512 | 1024 | 2048), ("no mem no phases", 62 | 128 | 256 | 512 | 1024 | 2048 | 4096)):
            bd.flags = fl_
            line += "  %s %.3f" % (name, timeit(fused) * 1e3)
        bd.flags = 0
        print(line, flush=True)
    a, b = timeit(three), timeit(fused)
    a2, b2 = timeit(three), timeit(fused)
    fl = 2.0 * N * H * W * (Cin * Cmid * 2 + 9 * Cmid * Cmid)
    by = N * H * W * Cin * 2 * 2.0
    same = bool(np.array_equal(o3.numpy(), of.numpy()))
    print("N%d %dx%d C%d mid%d: three %.3f / %.3f ms, fused %.3f / %.3f ms (%.2fx), fused = %.0f TF/s, %.2f TB/s of in+out; equal to three launches (auto tiles): %s" % (
        N, H, W, Cin, Cmid, a * 1e3, a2 * 1e3, b * 1e3, b2 * 1e3, min(a, a2) / min(b, b2), fl / min(b, b2) / 1e12, by / min(b, b2) / 1e12, same), flush=True)
