"""Dev tool: per-step cycle stamps of block 0 of the row-strip kernel (needs lib/libisegmi_trace.so, built with -DISEGMI_STRIP_TRACE; see
tools/strip_trace.sh).  Prints, per wave role, the mean cycles per step spent working / waiting at the barrier / waiting for loads."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
os.environ["ISEGMI_LIB"] = os.path.join(ROOT, "instancesegmentation-jittor_amd", "lib", "libisegmi_trace.so")
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/strip_trace.txt"
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 29
ZERO = len(sys.argv) > 3 and sys.argv[3] == "zero"
N, H, W, Cin, Cout = 8, 200, 336, 256, 256
rng = np.random.default_rng(0)
x = rng.standard_normal((N, H, W, Cin)).astype(np.float16)
if ZERO: x[:] = 0
w = (rng.standard_normal((Cout, 3, 3, Cin)) * 0.05).astype(np.float32)
d = _ffi.make_conv_desc(N, H, W, Cin, Cout, 3, 3, 1, 1, 1, tile)
dx = _ffi.DeviceBuffer.from_numpy(x); dw = _ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights_f16(d, w)); do = _ffi.DeviceBuffer((N, H, W, Cout), np.float16)
run = lambda: _ffi.check(_ffi.lib().isegmi_op_conv2d_f16(C.byref(d), dx.ptr, dw.ptr, None, None, None, do.ptr, 0, None))
for _ in range(5): run()
_ffi.sync()
os.environ["ISEGMI_STRIP_TRACE_DUMP"] = out
run(); _ffi.sync()
allrows = np.loadtxt(out, dtype=np.int64)
print("block 0: %d shader cycles in %d ticks of the 100 MHz counter -> %.3f GHz" % (allrows[0, 2], allrows[0, 3], allrows[0, 2] / max(1, allrows[0, 3]) / 10.0))
if os.environ.get("ISEGMI_STRIP_TRACE_LIGHT"): sys.exit(0)
rows = allrows[1:].reshape(-1, 64, 6)
nw = rows.shape[0]
nsteps = 36
print("tile %d: %d waves; cycles (s_memtime) per step, steps 2..%d" % (tile, nw, nsteps - 1))
for wv in range(nw):
    t = rows[wv, :, 2:].astype(np.int64) & 0xffffffff
    if wv < nw - 4:   # MFMA wave: slot 0 = arrive at barrier, 1 = leave
        arrive, leave = t[:nsteps, 0], t[:nsteps, 1]
        wait = (leave - arrive)[2:]
        work = (arrive[1:] - leave[:-1])[2:]
        epi = t[63, 1] - t[63, 0]
        print("  mfma wave %2d: work %6.0f  barrier wait %6.0f  (min/max work %d/%d)  epilogue %d  whole %d" % (wv, work.mean(), wait.mean(), work.min(), work.max(), epi, t[63, 1] - leave[0]))
    else:             # loader: 0 before waitcnt, 1 after it, 2 after the barrier, 3 after the issues
        a, b, c, dd = t[:nsteps, 0], t[:nsteps, 1], t[:nsteps, 2], t[:nsteps, 3]
        print("  loader wave %2d: load wait %6.0f  barrier wait %6.0f  issue %6.0f  (step %6.0f)" % (wv, (b - a)[2:].mean(), (c - b)[2:].mean(), (dd - c)[2:].mean(), np.diff(a)[2:].mean()))
t0 = rows[0, :nsteps, 3] & 0xffffffff
print("  step starts of wave 0 (leave barrier), deltas:", " ".join(str(int(v)) for v in np.diff(t0)))
