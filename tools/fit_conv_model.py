"""Fits the fp32 conv tile COST MODEL of csrc/conv_mfma.hip (conv2d_launch, tile == 0) to a per-layer sweep (tools/conv_tile_sweep.py ... canvases)
and reports how far the model's choice is from the per-layer best -- next to the hand-fitted ladder of rounds 1-2.

    python tools/fit_conv_model.py profiles/r03_conv_tile_sweep.txt [--emit]

Model (per candidate kernel c; ncu = compute units, nck = K / 32 chunks):
    blocks B_c = ceil(M / bm_c) * ceil(Cout / bn_c);   q = ceil(B_c / ncu)  blocks on the busiest CU
    T_c = f_c + max(q * (nck * m_c + em_c), ceil(q / occ_c) * (nck * l_c + el_c))        [us]
  m_c = matrix-pipe time of one chunk of one block, em_c = the pipe-side cost of a block's epilogue; l_c / el_c = the same two as LATENCIES,
  paid once per round of occ_c co-resident blocks when they cannot hide each other; f_c = launch + ring fill.  The hybrid launch (13) runs
  whole-CU multiples of 64x64 tiles on the v2 body and the left-over rows on 32x32 blocks:
  T_13 = f + q_main * (nck * m_a + em_a) + ceil(ntail / ncu / 8) * (nck * l_t + el_t).
--emit prints the C table for conv_mfma.hip."""
import math
import re
import sys

import numpy as np
from scipy.optimize import least_squares

NCU = 256
CAND = {4: (32, 32, 8), 5: (32, 32, 4), 6: (32, 64, 6), 10: (64, 64, 4), 12: (64, 64, 4)}


def parse(path):
    rows = []
    with open(path) as f:
        head = f.readline().split()
        tiles = [int(h[1:]) for h in head if re.fullmatch(r"t\d+", h) and h != "t64"]
        for ln in f:
            p = ln.split()
            if len(p) < 4 + len(tiles):
                continue
            m = re.search(r"\[M=(\d+) (\d)x\d/(\d)\]", ln)
            if not m:
                continue
            ts = dict(zip(tiles, (float(v) * 1e3 for v in p[3:3 + len(tiles)])))  # us
            rows.append(dict(M=int(m.group(1)), Cout=int(p[1]), K=int(p[2]), R=int(m.group(2)), t=ts, name=ln.split("  ")[-1].strip()))
    return tiles, rows


def blocks(M, Cout, bm, bn):
    return -(-M // bm) * -(-Cout // bn)


def model_time(c, prm, M, Cout, K):
    nck = K // 32
    if c == 13:
        f, ma, ea, lt, et = prm
        nt64, mt64 = -(-Cout // 64), -(-M // 64)
        main_mt = (mt64 * nt64 // NCU) * NCU // nt64
        if main_mt * 64 > M:
            main_mt = M // 64
        if main_mt <= 0 or main_mt >= mt64:
            return None
        qm = main_mt * nt64 / NCU
        ntail = -(-(M - main_mt * 64) // 32) * -(-Cout // 32)
        return f + qm * (nck * ma + ea) + math.ceil(ntail / NCU / 8) * (nck * lt + et)
    bm, bn, occ = CAND[c]
    f, m, em, l, el = prm
    q = math.ceil(blocks(M, Cout, bm, bn) / NCU)
    return f + max(q * (nck * m + em), math.ceil(q / occ) * (nck * l + el))


def fit(c, rows):
    data = [(r["M"], r["Cout"], r["K"], r["t"][c]) for r in rows if c in r["t"] and r["K"] % 32 == 0 and not (r["Cout"] <= 32 and c in (10, 12, 13, 6))]
    data = [d for d in data if model_time(c, (1.0,) * 5, d[0], d[1], d[2]) is not None]

    def res(x):
        return [math.log(max(model_time(c, np.abs(x), M, Co, K), 1e-3) / t) for M, Co, K, t in data]
    x0 = np.array([6.0, 0.25, 1.0, 0.5, 2.0])
    sol = least_squares(res, x0, loss="soft_l1", f_scale=0.1)
    prm = np.abs(sol.x)
    err = np.array(res(sol.x))
    return prm, float(np.sqrt(np.mean(err ** 2))), len(data)


def ladder(M, Cout, K, stem=False):
    """the rule of conv2d_launch (tile == 0), restated"""
    t64 = -(-M // 64) * -(-Cout // 64)
    nck = K // 32
    v2 = 12 if nck >= 72 else 10
    if t64 <= 176:
        return 5
    if nck <= 2 and Cout > 32:
        return v2
    if Cout <= 32 or t64 <= 480:
        return 4
    if 513 <= t64 and t64 % NCU != 0 and (t64 % NCU) * 100 <= t64 * 15:   # round 3: no upper limit (was 2600)
        return 13
    if t64 <= 512:
        return v2
    if t64 <= 640:
        return 6 if nck >= 32 else 4
    if t64 <= 1024:
        return v2
    if t64 <= 1100:
        return 6 if nck >= 32 else 4
    return v2


def main():
    tiles, rows = parse(sys.argv[1])
    rows = [r for r in rows if not (r["R"] == 7)]  # the stem has its own kernel
    params = {}
    for c in [t for t in tiles if t in CAND or t == 13]:
        params[c], rms, n = fit(c, rows)
        print("tile %2d: f %.2f us  m %.4f em %.3f  l %.4f el %.3f   (rms log error %.3f over %d layers)" % (c, *params[c], rms, n))

    def choose(r):
        best, bt = None, 1e30
        for c, prm in params.items():
            if r["Cout"] <= 32 and c not in (4, 5):
                continue
            t = model_time(c, prm, r["M"], r["Cout"], r["K"])
            if t is not None and t < bt:
                best, bt = c, t
        return best
    reg_m, reg_l, worst = [], [], []
    for r in rows:
        if r["K"] % 32:
            continue
        tb = min(r["t"].values())
        cm, cl = choose(r), ladder(r["M"], r["Cout"], r["K"])
        if cl == 13 and model_time(13, (1,) * 5, r["M"], r["Cout"], r["K"]) is None:
            cl = 10
        tm, tl = r["t"].get(cm), r["t"].get(cl)
        if tm is None or tl is None:
            continue
        reg_m.append((tm / tb - 1, r["t"][cm])); reg_l.append((tl / tb - 1, r["t"][cl]))
        if tm / tb - 1 > 0.03:
            worst.append((tm / tb - 1, cm, min(r["t"], key=r["t"].get), r["name"]))
    tot = sum(min(r["t"].values()) for r in rows if r["K"] % 32 == 0)
    print("layers %d; time if every layer ran its best tile %.1f us" % (len(reg_m), tot))
    for nm, reg in (("model ", reg_m), ("ladder", reg_l)):
        a = np.array([x[0] for x in reg]); t = np.array([x[1] for x in reg])
        print("%s: chosen-tile time / best-tile time: total %+.2f %%, mean per layer %+.2f %%, max %+.1f %%, layers over 3 %%: %d" % (
            nm, (t.sum() / tot - 1) * 100, a.mean() * 100, a.max() * 100, int((a > 0.03).sum())))
    for w in sorted(worst, reverse=True)[:12]:
        print("   model off by %+.1f %%: chose t%d, best t%d  %s" % (w[0] * 100, w[1], w[2], w[3]))
    if "--emit" in sys.argv:
        print("static const struct { int id, bm, bn, occ; float f, m, em, l, el; } FT[] = {")
        for c, prm in params.items():
            bm, bn, occ = CAND.get(c, (64, 64, 4))
            print("    {%d, %d, %d, %d, %.3ff, %.5ff, %.4ff, %.5ff, %.4ff}," % (c, bm, bn, occ, *prm))
        print("};")


if __name__ == "__main__":
    main()
