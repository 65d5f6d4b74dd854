"""Dev tool: bs=1 conv launches per forced fp32 tile (0 auto, 4: 32x32 on 16x16x4, 5: the same with four loader waves, 6: 32x64, 3: 64x64), us per launch."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
rng = np.random.default_rng(0)
TILES = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 4, 5, 6, 3]
SH = [(1, 138, 138, 64, 64, 3, 1), (1, 138, 138, 64, 256, 1, 0), (1, 69, 69, 128, 128, 3, 1), (1, 69, 69, 512, 128, 1, 0), (1, 35, 35, 256, 256, 3, 1), (1, 35, 35, 1024, 256, 1, 0), (1, 35, 35, 256, 1024, 1, 0),
      (1, 18, 18, 512, 512, 3, 1), (1, 18, 18, 2048, 512, 1, 0), (1, 18, 18, 512, 2048, 1, 0), (1, 69, 69, 256, 256, 3, 1)]
for (N, H, W, Cin, Cout, R, pad) in SH:
    w = (rng.standard_normal((Cout, R, R, Cin)) * 0.05).astype(np.float32)
    x = _ffi.DeviceBuffer.from_numpy(rng.standard_normal((N, H, W, Cin)).astype(np.float32)); o = _ffi.DeviceBuffer((N, H, W, Cout))
    line = "M=%-6d K=%-5d Cout=%-5d floor %5.1f us:" % (N * H * W, R * R * Cin, Cout, R * R * Cin / 4 * 33 / 2.4e3)
    for tile in TILES:
        d = _ffi.make_conv_desc(N, H, W, Cin, Cout, R, R, 1, pad, 1, tile)
        dw = _ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights(d, w))
        def run(): _ffi.check(_ffi.lib().isegmi_op_conv2d(C.byref(d), x.ptr, dw.ptr, None, None, None, o.ptr, None))
        try:
            for _ in range(3): run()
            _ffi.sync(); t0 = time.perf_counter()
            for _ in range(200): run()
            _ffi.sync(); line += "  t%d %5.1f" % (tile, (time.perf_counter() - t0) / 200 * 1e6)
        except Exception as e:
            line += "  t%d  n/a" % tile
    print(line, flush=True)
