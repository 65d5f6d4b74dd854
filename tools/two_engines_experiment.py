"""Experiment: two engines with half the batch each, enqueued alternately from one host thread, against one engine with the whole batch
(does a second independent kernel stream fill the CUs that a layer's last partial round of tiles leaves idle?).  python tools/two_engines_experiment.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
from isegmi.weights import yolact_state_dict
from isegmi.yolact import Yolact

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(0)
sd = yolact_state_dict(1234)
def make(bs):
    net = Yolact(sd, max_batch=bs)
    raw = rng.integers(0, 256, (bs, 550, 550, 3), dtype=np.uint8)
    pin = _ffi.PinnedBuffer(raw.shape, np.uint8); pin.array[...] = raw
    return net, pin, bs
def run(engs, steps):
    for net, pin, bs in engs: net.upload_u8_async(pin, bs, 550, 550, 0)
    for i in range(8):
        for net, pin, bs in engs:
            net.upload_u8_async(pin, bs, 550, 550, (i + 1) & 1); net.forward_device(bs, i & 1); net.postprocess_device(550, 550); net.mark_step()
    for net, _, _ in engs: net.sync()
    t0 = time.perf_counter()
    for i in range(steps):
        for net, pin, bs in engs:
            net.upload_u8_async(pin, bs, 550, 550, (i + 1) & 1); net.forward_device(bs, i & 1); net.postprocess_device(550, 550); net.mark_step(); net.wait_mark(1)
    for net, _, _ in engs: net.sync()
    el = time.perf_counter() - t0
    return sum(e[2] for e in engs) * steps / el
one = [make(8)]
print("one engine  bs=8      : %.1f img/s" % run(one, steps))
one[0][0].close()
two = [make(4), make(4)]
print("two engines bs=4 + 4  : %.1f img/s" % run(two, steps))
for e in two: e[0].close()
four = [make(8), make(8)]
print("two engines bs=8 + 8  : %.1f img/s" % run(four, steps))
