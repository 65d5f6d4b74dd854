import ctypes as C, os, sys, time
ROOT = "/root/repo"; sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
rng = np.random.default_rng(0)
for (N,H,W,Cin,Cout,R,pad) in [(1,4,8,32,32,1,0),(1,8,16,32,32,1,0),(1,35,35,32,32,1,0),(1,35,35,256,256,1,0),(1,35,35,256,1024,1,0),(1,138,138,64,64,1,0),(1,138,138,64,256,1,0)]:
    w = (rng.standard_normal((Cout, R, R, Cin)) * 0.05).astype(np.float32)
    x = _ffi.DeviceBuffer.from_numpy(rng.standard_normal((N, H, W, Cin)).astype(np.float32)); o = _ffi.DeviceBuffer((N, H, W, Cout))
    line="M=%d K=%d Cout=%d:"%(N*H*W,Cin*R*R,Cout)
    for tile in (4,3):
        d = _ffi.make_conv_desc(N, H, W, Cin, Cout, R, R, 1, pad, 1, tile)
        dw = _ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights(d, w))
        def run(): _ffi.check(_ffi.lib().isegmi_op_conv2d(C.byref(d), x.ptr, dw.ptr, None, None, None, o.ptr, None))
        for _ in range(3): run()
        for n in (200, 2000):
            _ffi.sync(); t0 = time.perf_counter()
            for _ in range(n): run()
            _ffi.sync(); line += "  t%d n%d %5.2f us" % (tile, n, (time.perf_counter() - t0) / n * 1e6)
    print(line, flush=True)
