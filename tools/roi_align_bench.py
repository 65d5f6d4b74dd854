"""Dev tool: the FPN heads' RoIAlign (7 x 7 with 1000 RoIs per image, or 14 x 14 with 100; P2-P5 of an 800 x 1344 canvas) -- the plain launch (one workgroup
per RoI, proposal order) against the table-driven one (isegmi_op_roi_prep, then one 128-byte channel slice of a RoI per workgroup) under row order, a random
order and roi_prep's, plus the prep pass alone.  argv: N [f16] [uniform|clustered] [mask]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
f16 = "f16" in sys.argv[2:]
clustered = "clustered" in sys.argv[2:]
K, Cc, PH = (100, 256, 14) if "mask" in sys.argv[2:] else (1000, 256, 7)
rng = np.random.default_rng(0)
dt = np.float16 if f16 else np.float32
shapes = [(200, 336), (100, 168), (50, 84), (25, 42)]
scales = [0.25, 0.125, 0.0625, 0.03125]
fb = [_ffi.DeviceBuffer.from_numpy(rng.standard_normal((N, h, w, Cc)).astype(dt)) for h, w in shapes]
def boxes(n):
    c = rng.uniform(0, 1, (n, 2)) * [1333, 800]
    if clustered:   # 20 objects, proposals scattered around them
        obj = rng.uniform(0.1, 0.9, (20, 2)) * [1333, 800]
        c = obj[rng.integers(0, 20, n)] + rng.normal(0, 30, (n, 2))
    wh = np.exp(rng.uniform(np.log(16), np.log(512), (n, 2)))
    b = np.concatenate([c - wh / 2, c + wh / 2], 1)
    return np.clip(b, 0, [1332, 799, 1332, 799]).astype(np.float32)
rois = np.stack([boxes(K) for _ in range(N)])
counts = np.full(N, K, np.int32)
ptrs = (C.c_void_p * 4)(*[b.ptr.value for b in fb])
Hs = (C.c_int32 * 4)(*[s[0] for s in shapes]); Ws = (C.c_int32 * 4)(*[s[1] for s in shapes]); sc = (C.c_float * 4)(*scales)
dr = _ffi.DeviceBuffer.from_numpy(rois); dcnt = _ffi.DeviceBuffer.from_numpy(counts)
do = _ffi.DeviceBuffer((N * K, PH, PH, Cc), dt); dord = _ffi.DeviceBuffer((N, K), np.int32); dtab = _ffi.DeviceBuffer((N * K, 4 * PH + 1, 4), np.int32)
L = _ffi.lib()
def timeit(run, n=50):
    for _ in range(5): run()
    _ffi.sync(); t0 = time.perf_counter()
    for _ in range(n): run()
    _ffi.sync(); return (time.perf_counter() - t0) / n * 1e6
def plain():
    if f16: _ffi.check(L.isegmi_op_roi_align_f16(ptrs, Hs, Ws, sc, 4, dr.ptr, dcnt.ptr, N, K, Cc, PH, PH, 2, 2, do.ptr, None))
    else: _ffi.check(L.isegmi_op_roi_align(ptrs, Hs, Ws, sc, 4, dr.ptr, dcnt.ptr, N, K, Cc, PH, PH, 2, 2, -1, do.ptr, None, None))
use_order = [True]
def ordered():
    fn = L.isegmi_op_roi_align_f16_ordered if f16 else L.isegmi_op_roi_align_ordered
    _ffi.check(fn(ptrs, Hs, Ws, sc, 4, dr.ptr, dcnt.ptr, dord.ptr if use_order[0] else None, dtab.ptr, N, K, Cc, PH, PH, 2, do.ptr, None))
def order():
    _ffi.check(L.isegmi_op_roi_prep(dr.ptr, dcnt.ptr, N, K, Hs, Ws, sc, 4, 2, Cc, PH, PH, 2 if f16 else 4, dord.ptr, dtab.ptr, None))
fp = sum(N * h * w * Cc for h, w in shapes) * np.dtype(dt).itemsize + N * K * PH * PH * Cc * np.dtype(dt).itemsize
print("N %d %s %s: algorithmic %.0f MB" % (N, "f16" if f16 else "f32", "clustered" if clustered else "uniform", fp / 1e6))
plain(); _ffi.sync(); ref = do.numpy().copy()
t = timeit(plain); print("plain                      %7.1f us  %.0f GB/s" % (t, fp / t / 1e3))
print("prep pass                  %7.1f us" % timeit(order))
for name, o in (("row order", None), ("random", rng.permutation(N * K).astype(np.int32)), ("roi_prep order", 0)):
    order(); _ffi.sync()
    use_order[0] = o is not None
    if use_order[0] and not isinstance(o, int): _ffi.check(L.isegmi_h2d(dord.ptr, o.ctypes.data_as(C.c_void_p), C.c_int64(o.nbytes)))
    ordered(); _ffi.sync()
    assert np.array_equal(do.numpy(), ref)
    t = timeit(ordered); print("table, %-18s  %7.1f us  %.0f GB/s" % (name, t, fp / t / 1e3), flush=True)
