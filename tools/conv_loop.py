"""Dev tool: run one fp16 conv (tile, shape as conv_f16_bench) in a loop for ~12 s (for tools/power_probe.sh).  argv: tile zero|rand|relu"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
tile = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "rand"
N, H, W, Cin, Cout = 8, 200, 336, 256, 256
rng = np.random.default_rng(0)
x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
if mode == "zero": x[:] = 0
if mode == "relu": x = np.maximum(x, 0)
w = (rng.standard_normal((Cout, 3, 3, Cin)) * 0.05).astype(np.float32)
d = _ffi.make_conv_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1, tile)
dx = _ffi.DeviceBuffer.from_numpy(x); dw = _ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights_f16(d, w)); do = _ffi.DeviceBuffer((N, H, W, Cout), np.float16)
run = lambda: _ffi.check(_ffi.lib().isegmi_op_conv2d_f16(C.byref(d), dx.ptr, dw.ptr, None, None, None, do.ptr, 0, None))
for _ in range(5): run()
_ffi.sync()
t_end = time.time() + 12
n = 0; t0 = time.perf_counter()
while time.time() < t_end:
    for _ in range(200): run()
    _ffi.sync(); n += 200
dt = (time.perf_counter() - t0) / n
print("tile %d %s: %.3f ms  %.0f TF/s" % (tile, mode, dt * 1e3, 2.0 * N * H * W * Cout * 9 * Cin / dt / 1e12))
