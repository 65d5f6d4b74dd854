"""Stage timings at a given batch size (dev tool): python tools/stage_times.py [yolact|maskrcnn] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np, time
model = sys.argv[1] if len(sys.argv) > 1 else "yolact"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(1)
if model == "yolact":
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, fast_base_transform
    net = Yolact(yolact_state_dict(1234), max_batch=bs)
    net.upload(fast_base_transform(rng.uniform(0, 255, (bs, 550, 550, 3)).astype(np.float32)))
    step = lambda: (net.forward_device(bs), net.postprocess_device(550, 550))
else:
    from isegmi.weights import maskrcnn_state_dict
    from isegmi.maskrcnn import MaskRCNN, prepare_images
    x, hw = prepare_images([rng.uniform(0, 255, (800, 1333, 3)).astype(np.float32) for _ in range(bs)])
    net = MaskRCNN(maskrcnn_state_dict(1234), x.shape[1], x.shape[2], max_batch=bs)
    net.upload(x, hw)
    step = lambda: (net.forward_device(bs), net.paste_device(800, 1333))
for _ in range(3): step()
net.sync()
t0 = time.perf_counter()
for _ in range(10): step()
net.sync()
print("wall ms/step %.3f" % ((time.perf_counter() - t0) * 100))
t0 = time.perf_counter()
for _ in range(10): step()
tl = (time.perf_counter() - t0) * 100
net.sync()
print("host launch ms/step %.3f" % tl)
net.set_param("timing", 1.0); step(); net.sync()
tm = net.timings(); print("stages", {k: round(v, 3) for k, v in tm}, "sum %.3f" % sum(v for _, v in tm))
