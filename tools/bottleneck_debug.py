import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
from test_bottleneck_f16_gpu import _weights, _three_launches
Cin, Cmid, N, H, W = 256, 64, 4, 64, 96
for mode in ("all", "one_channel", "16ch_one_mfma", "64ch_one_chunk", "128ch", "all_scale1"):
    rng = np.random.default_rng(7)
    x = np.maximum(rng.standard_normal((N, H, W, Cin)), 0).astype(np.float16)
    x[..., :Cmid] = 0
    if mode == "one_channel": x[..., :100] = 0; x[..., 101:] = 0
    if mode == "16ch_one_mfma": x[..., :128] = 0; x[..., 144:] = 0
    if mode == "64ch_one_chunk": x[..., :128] = 0; x[..., 192:] = 0
    if mode == "128ch": x[..., :128] = 0
    w1, sb1, w2, sb2, w3, sb3 = _weights(rng, Cin, Cmid)
    if mode == "all_scale1": sb1 = (np.ones(Cmid, np.float32), np.zeros(Cmid, np.float32))
    w2 = np.zeros_like(w2); w2[np.arange(Cmid), 1, 1, np.arange(Cmid)] = 1.0; sb2 = (np.ones(Cmid, np.float32), np.zeros(Cmid, np.float32))
    w3 = np.zeros_like(w3); w3[np.arange(Cmid), 0, 0, np.arange(Cmid)] = 1.0; sb3 = (np.ones(Cin, np.float32), np.zeros(Cin, np.float32))
    t1ref = _ffi.conv2d_f16(x, w1, 1, 0, sb1[0], sb1[1], None, 1, 4)
    got = _ffi.bottleneck_f16(x, w1, sb1, w2, sb2, w3, sb3)[..., :Cmid]
    print("%-16s t1 elements differing: %d of %d (nonzero in ref: %d)" % (mode, int((got != t1ref).sum()), got.size, int((t1ref != 0).sum())), flush=True)
