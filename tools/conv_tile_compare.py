"""Per-layer conv time under forced tiles, inside the model (dev tool): python tools/conv_tile_compare.py [yolact|maskrcnn] [batch] [tiles=0,13,14,6,4,10]
tile 0 = the launcher's own rule (run with conv_groups 0, so that every layer is a launch of its own like the forced ones).  One line per layer
shape (layers of equal shape averaged): ms per tile, the rule's choice against the best forced tile."""
import ctypes as C, sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
model = sys.argv[1] if len(sys.argv) > 1 else "yolact"
bs = int(sys.argv[2]) if len(sys.argv) > 2 else (8 if model == "yolact" else 2)
tiles = [int(t) for t in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 13, 14, 6, 4, 10]
rng = np.random.default_rng(1)
if model == "yolact":
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact, fast_base_transform
    net = Yolact(yolact_state_dict(1234), max_batch=bs)
    net.upload(fast_base_transform(rng.uniform(0, 255, (bs, 550, 550, 3)).astype(np.float32)))
else:
    from isegmi.weights import maskrcnn_state_dict
    from isegmi.maskrcnn import MaskRCNN, prepare_images
    x, hw = prepare_images([rng.uniform(0, 255, (800, 1333, 3)).astype(np.float32) for _ in range(bs)])
    net = MaskRCNN(maskrcnn_state_dict(1234), x.shape[1], x.shape[2], max_batch=bs)
    net.upload(x, hw)
step = lambda: net.forward_device(bs)
net.set_param("multi_stream", 0.0); net.set_param("conv_groups", 0.0)
res = collections.defaultdict(dict)
R = 5
for rep in range(2):            # two passes over the tiles: the second one is reported (clocks settled)
    for t in tiles:
        net.set_param("conv_tile", float(t))
        for _ in range(2): step()
        net.sync(); net.set_param("conv_timing", 1.0)
        f, m, l = C.c_double(), C.c_double(), C.c_int64()
        _ffi.lib().isegmi_engine_conv_stats(net._h, C.byref(f), C.byref(m), C.byref(l))
        buf = C.create_string_buffer(1 << 18); _ffi.lib().isegmi_engine_conv_report(net._h, buf, 1 << 18)
        for _ in range(R): step()
        net.sync()
        _ffi.check(_ffi.lib().isegmi_engine_conv_report(net._h, buf, 1 << 18))
        net.set_param("conv_timing", 0.0)
        acc = collections.defaultdict(list)
        for r in (q.split("\t") for q in buf.value.decode().strip().split("\n")):
            shape = r[0][r[0].index("["):]
            acc[shape].append(float(r[2]) / R)
        for k, v in acc.items(): res[k][t] = (sum(v), len(v))
tot = {t: sum(res[k][t][0] for k in res if t in res[k]) for t in tiles}
print("total ms/step:", {t: round(v, 3) for t, v in tot.items()})
gain = 0.0
for k in sorted(res, key=lambda k: -res[k][tiles[0]][0]):
    r = res[k]
    best = min((t for t in tiles if t in r), key=lambda t: r[t][0])
    d = r[tiles[0]][0] - r[best][0]
    gain += d
    print("%-52s x%-2d " % (k, r[tiles[0]][1]) + "  ".join("t%d %.3f" % (t, r[t][0] / r[t][1]) for t in tiles if t in r) + "   best t%d (%+.3f ms/step)" % (best, -d))
print("sum over layers of (rule - best forced): %.3f ms/step" % gain)
