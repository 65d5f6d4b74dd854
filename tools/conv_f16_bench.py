"""Micro-benchmark of the fp16 / fp32 conv kernels on Mask R-CNN shapes (dev tool)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
rng = np.random.default_rng(0)
shapes = [(8, 200, 336, 256, 256, 3, 1, 1), (8, 100, 168, 256, 256, 3, 1, 1), (8, 200, 336, 64, 256, 1, 1, 0), (8, 200, 336, 256, 64, 1, 1, 0),
          (8, 200, 336, 64, 64, 3, 1, 1), (8, 100, 168, 128, 512, 1, 1, 0), (8, 100, 168, 128, 128, 3, 1, 1),
          (8, 50, 84, 1024, 256, 1, 1, 0), (8, 50, 84, 256, 256, 3, 1, 1), (8, 50, 84, 256, 1024, 1, 1, 0), (8, 25, 42, 512, 512, 3, 1, 1),
          (8, 25, 42, 512, 2048, 1, 1, 0), (800, 14, 14, 256, 256, 3, 1, 1), (8000, 7, 7, 256, 1024, 7, 1, 0)]
TILES = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 3, 4, 5, 9, 10, 11, 12, 13, 14, 16, 26, 27, 28]
WITH_RES = len(sys.argv) > 2 and sys.argv[2] == "res"
ONLY = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
for si, (N, H, W, Cin, Cout, R, st, pad) in enumerate(shapes):
    if ONLY is not None and si not in ONLY:
        continue
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
    w = (rng.standard_normal((Cout, R, R, Cin)) * 0.05).astype(np.float32)
    ho, wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    fl = 2.0 * N * ho * wo * Cout * R * R * Cin
    line = "N%d %dx%d Cin%d Cout%d %dx%d/%d  %.1f GF:" % (N, H, W, Cin, Cout, R, R, st, fl / 1e9)
    for tile in TILES:
        if 26 <= (tile & 255) <= 31 and not (R == 3 and st == 1 and pad == 1):
            continue
        d = _ffi.make_conv_desc(N, H, W, Cin, Cout, R, R, st, pad, 1, tile)
        dx = _ffi.DeviceBuffer.from_numpy(x); dw = _ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights_f16(d, w)); do = _ffi.DeviceBuffer((N, ho, wo, Cout), np.float16); dr = _ffi.DeviceBuffer((N, ho, wo, Cout), np.float16)
        run = lambda: _ffi.check(_ffi.lib().isegmi_op_conv2d_f16(C.byref(d), dx.ptr, dw.ptr, None, None, dr.ptr if WITH_RES else None, do.ptr, 0, None))
        for _ in range(3): run()
        _ffi.sync(); t0 = time.perf_counter()
        for _ in range(20): run()
        _ffi.sync(); dt = (time.perf_counter() - t0) / 20
        line += "  t%d %.3f ms %.0f" % (tile, dt * 1e3, fl / dt / 1e12)
    print(line, flush=True)
