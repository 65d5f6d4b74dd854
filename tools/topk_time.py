import os, sys, time, ctypes as C
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
rng = np.random.default_rng(0)
for rows, n, k in [(8, 201600, 1000), (33, 201600, 1000), (8, 50400, 1000), (33, 50400, 1000), (8, 32768, 1000), (33, 32768, 1000), (8, 12600, 1000)]:  # rows = 33: one level (one block per row)
    keys = _ffi.DeviceBuffer.from_numpy(rng.uniform(0, 1, (rows, n)).astype(np.float32))
    v = _ffi.DeviceBuffer((rows, k)); i = _ffi.DeviceBuffer((rows, k), np.int32); c = _ffi.DeviceBuffer((rows,), np.int32)
    run = lambda: _ffi.check(_ffi.lib().isegmi_op_topk(keys.ptr, C.c_int64(n), rows, n, k, None, 1, v.ptr, i.ptr, c.ptr, None))
    for _ in range(3): run()
    _ffi.sync(); t0 = time.perf_counter()
    for _ in range(50): run()
    _ffi.sync(); print(rows, n, k, "%.1f us" % ((time.perf_counter() - t0) / 50 * 1e6), flush=True)
