"""Dev tool: the split-K tile (15) against the tiles the ladder picks (0) and the forced small-grid tiles (4, 5) on the bs = 1 backbone shapes; us per launch."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
rng = np.random.default_rng(0)
shapes = [(1, 35, 35, 256, 256, 3, 1), (1, 35, 35, 1024, 256, 1, 1), (1, 18, 18, 512, 512, 3, 1), (1, 18, 18, 2048, 512, 1, 1), (1, 69, 69, 128, 128, 3, 1),
          (1, 25, 42, 512, 512, 3, 1), (1, 25, 42, 2048, 512, 1, 1), (1, 35, 35, 256, 1024, 1, 1)]
for (N, H, W, Cin, Cout, R, st) in shapes:
    pad = R // 2
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32); w = (rng.standard_normal((Cout, R, R, Cin)) * 0.05).astype(np.float32)
    line = "N%d %dx%d Cin%d Cout%d %dx%d:" % (N, H, W, Cin, Cout, R, R)
    for tile in (0, 4, 5, 15):
        d = _ffi.make_conv_desc(N, H, W, Cin, Cout, R, R, st, pad, 1, tile)
        ho, wo = _ffi.conv_out_hw(d)
        dx = _ffi.DeviceBuffer.from_numpy(x); dw = _ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights(d, w)); do = _ffi.DeviceBuffer((N, ho, wo, Cout))
        run = lambda: _ffi.op_conv2d(d, dx, dw, None, None, None, do)
        for _ in range(5): run()
        _ffi.sync(); t0 = time.perf_counter()
        for _ in range(50): run()
        _ffi.sync(); dt = (time.perf_counter() - t0) / 50
        line += "  t%d %.1f us" % (tile, dt * 1e6)
    print(line + "   qualifies=%d" % _ffi.lib().isegmi_conv_split_qualifies(C.byref(d)), flush=True)
