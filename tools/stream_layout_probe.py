"""Stream -> hardware-queue placement against throughput (dev tool): k foreign streams are created before the engine; prints the queue class of
each of the engine's ten streams (isegmi_engine_stream_layout) and the Yolact bs=8 img/s.   python tools/stream_layout_probe.py K [K ...]"""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
if len(sys.argv) > 2 or (len(sys.argv) == 2 and "," in sys.argv[1]):
    ks = sys.argv[1].split(",") if "," in sys.argv[1] else sys.argv[1:]
    rc = 0
    for k in ks:   # one fresh process per k: the placement is a property of the process's history
        r = subprocess.run([sys.executable, os.path.abspath(__file__), str(k)], capture_output=True, text=True)
        print(r.stdout.strip(), flush=True)
        if r.returncode != 0:   # a failed child is this tool's failure, with its reason
            sys.stderr.write("k=%s: child exited with %d\n%s\n" % (k, r.returncode, r.stderr[-2000:]))
            rc = rc or r.returncode
    sys.exit(rc)
import numpy as np
from isegmi import _ffi
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_ffi.set_device(0)
hip = C.CDLL("libamdhip64.so")
foreign = []
for i in range(k):
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    foreign.append(st)
from isegmi.weights import yolact_state_dict
from isegmi.yolact import Yolact
bs = 8
net = Yolact(yolact_state_dict(1234), max_batch=bs)
q = (C.c_int32 * 10)()
_ffi.check(_ffi.lib().isegmi_engine_stream_layout(net._h, q, 10))
rng = np.random.default_rng(0)
raw = rng.integers(0, 256, (bs, 550, 550, 3), dtype=np.uint8)
pin = _ffi.PinnedBuffer(raw.shape, np.uint8); pin.array[...] = raw
net.upload_u8_async(pin, bs, 550, 550, 0)
def loop(n):
    for i in range(n):
        net.upload_u8_async(pin, bs, 550, 550, (i + 1) & 1); net.forward_device(bs, i & 1); net.postprocess_device(550, 550); net.mark_step(); net.wait_mark(1)
loop(10); net.sync()
t0 = time.perf_counter(); loop(50); net.sync(); el = time.perf_counter() - t0
names = ["main", "side0", "side1", "side2", "tail", "heads", "hs0", "hs1", "hs2", "copy"]
print("k=%d  %.1f img/s  queues: %s" % (k, bs * 50 / el, " ".join("%s:%d" % (n, c) for n, c in zip(names, q))))
