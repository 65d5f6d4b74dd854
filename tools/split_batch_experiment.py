"""Experiment (dev tool): does running a batch as S independent sub-batches on S engines (own streams and buffers, enqueued back to back from
one host thread) beat one engine at the full batch?  Smaller launches co-run on the chip, so the tail of one layer overlaps the head of another.
   python tools/split_batch_experiment.py [yolact|maskrcnn] [batch] [splits=1,2,4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
model = sys.argv[1] if len(sys.argv) > 1 else "yolact"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
splits = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4]
rng = np.random.default_rng(1)
for S in splits:
    sub = bs // S
    nets = []
    if model == "yolact":
        from isegmi.weights import yolact_state_dict
        from isegmi.yolact import Yolact, fast_base_transform
        sd = yolact_state_dict(1234)
        for i in range(S):
            net = Yolact(sd, max_batch=sub)
            net.upload(fast_base_transform(rng.uniform(0, 255, (sub, 550, 550, 3)).astype(np.float32)))
            nets.append(net)
    else:
        from isegmi.weights import maskrcnn_state_dict
        from isegmi.maskrcnn import MaskRCNN, MaskRCNNConfig, prepare_images
        sd = maskrcnn_state_dict(1234, 50)
        for i in range(S):
            x, hw = prepare_images([rng.uniform(0, 255, (800, 1333, 3)).astype(np.float32) for _ in range(sub)])
            net = MaskRCNN(sd, x.shape[1], x.shape[2], cfg=MaskRCNNConfig(depth=50), max_batch=sub)
            net.upload(x, hw)
            nets.append(net)
    for _ in range(3):
        for net in nets: net.forward_device(sub)
    for net in nets: net.sync()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        for net in nets: net.forward_device(sub)
    for net in nets: net.sync()
    dt = (time.perf_counter() - t0) / K
    print("%s batch %d as %d x %d: %.3f ms per batch, %.1f img/s" % (model, bs, S, sub, dt * 1e3, bs / dt), flush=True)
    for net in nets: net.close()
