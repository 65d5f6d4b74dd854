"""Micro-benchmark of the fp32 conv kernels on the ResNet-50 body shapes of both models (dev tool):
   python tools/conv_f32_bench.py [tiles=0,3,4,6] [set=y8|m2|all] [rot=3]
Each shape is timed with HIP-event-free wall clock over back-to-back launches that ROTATE over `rot` buffer sets, so that a
layer whose tensors exceed the 256 MiB Infinity Cache is not served from it (inside the model the producer's output is equally
cold).  Prints ms, TF/s and the algorithmic HBM GB/s (input + weights + residual + output once)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, "instancesegmentation-jittor_amd")]
import numpy as np
from isegmi import _ffi
_ffi.set_device(0)
rng = np.random.default_rng(0)


def body(N, H, W):
    """(N, H, W, Cin, Cout, R, stride, pad, residual) of one bottleneck per stage at stem-output size H x W (after the max-pool)."""
    out = []
    h, w, cin = H, W, 64
    for li, mid in enumerate((64, 128, 256, 512)):
        st = 1 if li == 0 else 2
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        # torchvision-style (stride on the 3x3) geometry; the Detectron style (stride on the first 1x1) has the same GEMM sizes
        # from the second block on, which is what dominates
        out += [(N, ho, wo, mid * 4, mid, 1, 1, 0, 0), (N, ho, wo, mid, mid, 3, 1, 1, 0), (N, ho, wo, mid, mid * 4, 1, 1, 0, 1)]
        h, w, cin = ho, wo, mid * 4
    return out


SETS = {"y8": body(8, 138, 138), "m2": body(2, 200, 336), "y4": body(4, 138, 138)[:6], "m1": body(1, 200, 336)[:6],
        # the large layers outside the ResNet body: Yolact P3 3x3s / protonet at 138^2 / fused 351-wide head; Mask R-CNN FPN + RPN 3x3s at
        # P2 / P3, mask-head 3x3 on 200 RoIs, FC6 as a 7x7 valid conv on 2000 RoIs
        # res4 / res5 of Yolact at bs 8: grids of one to three 64x64 rounds (the 16x16x4 kernels' territory)
        "mid": [(8, 35, 35, 256, 256, 3, 1, 1, 0), (8, 35, 35, 1024, 256, 1, 1, 0, 0), (8, 35, 35, 256, 1024, 1, 1, 0, 1), (8, 18, 18, 512, 512, 3, 1, 1, 0),
                (2, 50, 84, 256, 256, 3, 1, 1, 0), (2, 50, 84, 1024, 256, 1, 1, 0, 0)],
        "big": [(8, 69, 69, 256, 256, 3, 1, 1, 0), (8, 138, 138, 256, 256, 3, 1, 1, 0), (8, 69, 69, 256, 351, 3, 1, 1, 0), (8, 35, 35, 256, 351, 3, 1, 1, 0),
                (2, 200, 336, 256, 256, 3, 1, 1, 0), (2, 100, 168, 256, 256, 3, 1, 1, 0), (200, 14, 14, 256, 256, 3, 1, 1, 0),
                (2000, 7, 7, 256, 1024, 7, 1, 0, 0), (2, 200, 336, 256, 256, 1, 1, 0, 0), (2, 200, 336, 64, 256, 1, 1, 0, 1)]}
TILES = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 3, 4, 6]
which = sys.argv[2] if len(sys.argv) > 2 else "all"
ROT = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for name, shapes in SETS.items():
    if which not in ("all", name):
        continue
    for (N, H, W, Cin, Cout, R, st, pad, res) in shapes:
        ho, wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
        M = N * ho * wo
        fl = 2.0 * M * Cout * R * R * Cin
        byt = 4.0 * (N * H * W * Cin + Cout * R * R * Cin + M * Cout * (2 if res else 1))
        w = (rng.standard_normal((Cout, R, R, Cin)) * 0.05).astype(np.float32)
        d0 = _ffi.make_conv_desc(N, H, W, Cin, Cout, R, R, st, pad, 1, 0)
        dw = _ffi.DeviceBuffer.from_numpy(_ffi.pack_conv_weights(d0, w))
        xs = [_ffi.DeviceBuffer.from_numpy(rng.standard_normal((N, H, W, Cin)).astype(np.float32)) for _ in range(ROT)]
        os_ = [_ffi.DeviceBuffer((N, ho, wo, Cout)) for _ in range(ROT)]
        rs = [_ffi.DeviceBuffer.from_numpy(rng.standard_normal((N, ho, wo, Cout)).astype(np.float32)) for _ in range(ROT)] if res else None
        line = "%s M=%-6d K=%-4d Cout=%-4d %dx%d%s %6.2f GF %6.1f MB:" % (name, M, R * R * Cin, Cout, R, R, "+res" if res else "    ", fl / 1e9, byt / 1e6)
        for tile in TILES:
            d = _ffi.make_conv_desc(N, H, W, Cin, Cout, R, R, st, pad, 1, tile)
            def run(i):
                _ffi.check(_ffi.lib().isegmi_op_conv2d(C.byref(d), xs[i % ROT].ptr, dw.ptr, None, None, rs[i % ROT].ptr if res else None, os_[i % ROT].ptr, None))
            try:
                for i in range(3): run(i)
            except _ffi.IsegmiError:
                line += "  t%d n/a" % tile
                continue
            _ffi.sync(); t0 = time.perf_counter()
            REP = 30
            for i in range(REP): run(i)
            _ffi.sync(); dt = (time.perf_counter() - t0) / REP
            line += "  t%d %.3f ms %5.1f TF %4.2f TB/s" % (tile, dt * 1e3, fl / dt / 1e12, byt / dt / 1e12)
        print(line, flush=True)
        for b in xs + os_ + (rs or []) + [dw]:
            b.free()
