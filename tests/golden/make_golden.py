"""Generates tests/golden/*.npz: seeded inputs + expected outputs of the hot-path ops.

PROVENANCE: the reference (Jittor/InstanceSegmentation-jittor) holds no runnable code, tests or
golden vectors for this path (SURVEY.md sections 0, 4, 8c), so these vectors come from THIS repo's CPU
oracle (oracle/ora_ops.c, SURVEY App. A revision 2026-10-03) plus hand-derived known-answer
cases.  They certify HIP == CPU restatement, not == Jittor ("parity unpinned").
Run:  python tests/golden/make_golden.py [name ...]   (no names = all fixtures)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ora  # noqa: E402


def boxes(rng, n, W, H):
    c = rng.uniform(0, 1, (n, 2)) * [W, H]
    c[n // 2:] = c[: n - n // 2] + rng.normal(0, 5, (n - n // 2, 2))
    wh = np.exp(rng.uniform(np.log(8), np.log(200), (n, 2)))
    return np.clip(np.concatenate([c - wh / 2, c + wh / 2], 1), 0, [W - 1, H - 1, W - 1, H - 1]).astype(np.float32)


def main():
    rng = np.random.default_rng(20261003)
    g = {}
    # conv 3x3 s1 with BN+residual+relu, and 1x1 s2
    x = rng.standard_normal((1, 9, 11, 32)).astype(np.float32); w = (rng.standard_normal((40, 3, 3, 32)) * 0.1).astype(np.float32)
    sc = rng.uniform(0.5, 1.5, 40).astype(np.float32); sh = (rng.standard_normal(40) * 0.1).astype(np.float32)
    res = rng.standard_normal((1, 9, 11, 40)).astype(np.float32)
    g["conv"] = dict(x=x, w=w, scale=sc, shift=sh, res=res, y=ora.conv2d(x, w, 1, 1, sc, sh, res, 1))
    w1 = (rng.standard_normal((16, 1, 1, 32)) * 0.2).astype(np.float32)
    g["conv1x1s2"] = dict(x=x, w=w1, y=ora.conv2d(x, w1, 2, 0, None, None, None, 0))
    # detmath
    v = np.concatenate([np.linspace(-20, 20, 801), rng.standard_normal(200) * 3]).astype(np.float32)
    g["detmath"] = dict(x=v, exp=ora.map_f32(v, 0), sigmoid=ora.map_f32(v, 1), tanh=ora.map_f32(v, 2), log2=ora.map_f32(np.abs(v) + 1e-3, 3))
    # nms
    b = boxes(rng, 120, 400, 300); s = rng.uniform(0, 1, 120).astype(np.float32)
    g["nms"] = dict(boxes=b, scores=s, keep_gt=ora.nms(b, s, 0.5, 1, 0), keep_ge=ora.nms(b, s, 0.5, 1, 1), keep_noplus=ora.nms(b, s, 0.5, 0, 0))
    # roi align on one level
    f = rng.standard_normal((1, 25, 42, 8)).astype(np.float32)
    r = np.concatenate([np.zeros((30, 1), np.float32), boxes(rng, 30, 336, 200)], 1)
    g["roi_align"] = dict(feat=f, rois=r, out7=ora.roi_align(f, r, 0.125, 7, 7, 2), levels=ora.level_map(r[:, 1:]))
    # yolact detect + masks
    P = 600
    conf = rng.standard_normal((P, 81)).astype(np.float32); conf[:, 0] += 3; hot = rng.integers(0, P, 60); conf[hot, rng.integers(1, 81, 60)] += 7
    pri = np.concatenate([rng.uniform(0.1, 0.9, (P, 2)), rng.uniform(0.05, 0.4, (P, 2))], 1).astype(np.float32); pri[1::2, :2] = pri[::2, :2] + 0.01
    loc = (rng.standard_normal((P, 4)) * 0.5).astype(np.float32); msk = np.tanh(rng.standard_normal((P, 32))).astype(np.float32)
    bx = ora.yolact_decode(loc, pri)
    d = ora.yolact_detect(ora.softmax(conf), bx, msk)
    proto = np.maximum(rng.standard_normal((24, 24, 32)), 0).astype(np.float32)
    mm, ib = ora.yolact_masks(proto, d["mask"], d["box"], 50, 60)
    g["yolact"] = dict(conf=conf, loc=loc, mask=msk, priors=pri, boxes=bx, det_prior=d["prior"], det_cls=d["cls"], det_score=d["score"],
                       proto=proto, masks=np.packbits(mm), masks_shape=np.array(mm.shape), int_boxes=ib)
    # paste
    m28 = rng.uniform(0, 1, (5, 28, 28)).astype(np.float32)
    pb = np.array([[10.2, 20.7, 80.1, 90.9], [-12.5, -3.0, 30.0, 44.4], [100, 50, 159.5, 119.2], [40, 40, 40.3, 40.2], [0, 0, 159, 119]], np.float32)
    g["paste"] = dict(masks=m28, boxes=pb, out=np.packbits(ora.paste_masks(m28, pb, 120, 160)))
    # DCNv2 sampling stage (YOLACT++ backbones): own generator so that the fixtures above stay byte-stable
    r2 = np.random.default_rng(20261004)
    xd = r2.standard_normal((1, 9, 11, 8)).astype(np.float32)
    omd = r2.normal(0, 2.0, (1, 5, 6, 27)).astype(np.float32); omd[0, 0, :, :18] = np.round(omd[0, 0, :, :18])
    g["deform"] = dict(x=xd, om=omd, col=ora.deform_im2col(xd, omd, 3, 3, 2, 1, 1))
    # front end (M1 / Y1): uint8 BGR in, the networks' fp32 input out; own generator (the fixtures above stay byte-stable)
    r3 = np.random.default_rng(20261005)
    yi = r3.integers(0, 256, (2, 37, 53, 3), dtype=np.uint8); yi[0, :2, :3] = 0; yi[1, -2:, -2:] = 255
    mi = [r3.integers(0, 256, (45, 70, 3), dtype=np.uint8), r3.integers(0, 256, (61, 40, 3), dtype=np.uint8)]
    mo, mhw = ora.to_image_list(mi)
    g["frontend"] = dict(y_in=yi, y_out64=ora.fast_base_transform(yi, 64), y_out24=ora.fast_base_transform(yi, 24), y_dark=ora.fast_base_transform(yi, 64, darknet=True),
                         m_in0=mi[0], m_in1=mi[1], m_out=mo, m_hw=mhw)
    only = sys.argv[1:]
    for k, v in g.items():
        if only and k not in only:
            continue
        np.savez_compressed(os.path.join(HERE, k + ".npz"), **v)
        print(k, {a: getattr(b, "shape", None) for a, b in v.items()})


if __name__ == "__main__":
    main()
