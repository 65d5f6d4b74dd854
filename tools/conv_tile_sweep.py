"""Per-layer tile comparison on the GPU (dev tool): runs tools/conv_report.py for every (model, batch) with each forced tile and merges
the per-layer times:  python tools/conv_tile_sweep.py 3,4,5,6,12 [f16] > gpurun_out/sweep.txt
Columns: t64 (number of 64x64 tiles), Cout, K, ms per tile id, best, source (cry8 = Yolact bs 8, crm2 = Mask R-CNN bs 2 ...), layer."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "3,4,5,6,12").split(",")]
runs = [("yolact", 1), ("yolact", 8), ("maskrcnn", 1), ("maskrcnn", 2)]
extra = []
canvases = {}
if len(sys.argv) > 2 and sys.argv[2] == "canvases":  # fp32, plus the padded canvases COCODemo produces on COCO images (800x1088, 800x1216, 608x800 ...)
    sizes = [(800, 1088), (800, 1216), (608, 800), (800, 800), (1216, 800)]
    runs += [("maskrcnn", 100 + 10 * i + b) for i in range(len(sizes)) for b in (1, 2)]
    canvases = {100 + 10 * i + b: (b, sizes[i]) for i in range(len(sizes)) for b in (1, 2)}
if len(sys.argv) > 2 and sys.argv[2] == "f16":  # the fp16 family: R101 bs 8 (configs[4]) and R50 bs 2
    runs = [("maskrcnn", 8), ("maskrcnn", 2)]
    extra = {8: ["fp16", "101"], 2: ["fp16", "50"]}
rows = {}
for model, bs in runs:
    for t in tiles:
        if bs in canvases:
            b, (ih, iw) = canvases[bs]
            args = [str(b), str(t), model, "fp32", "50", str(ih), str(iw)]
        else:
            args = [str(bs), str(t), model] + (extra[bs] if extra else [])
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv_report.py")] + args, capture_output=True, text=True, timeout=600).stdout
        print("swept", model, bs, "tile", t, file=sys.stderr, flush=True)
        for ln in out.splitlines():
            m = re.match(r"(\S+) \[M=(\d+) K=(\d+) Cout=(\d+) (\d)x\d/(\d)\]\s+([\d.]+) GF\s+([\d.]+) ms", ln)
            if not m:
                continue
            name, M, K, Cout, R, st, gf, ms = m.groups()
            key = ("cr%s%d" % (model[0], bs), name, int(M), int(K), int(Cout), "%sx%s/%s" % (R, R, st))
            rows.setdefault(key, {})[t] = float(ms)
seen = set()
lines = []
for (src, name, M, K, Cout, geo), ts in rows.items():
    sig = (M, K, Cout, geo)
    if sig in seen or len(ts) < len(tiles):
        continue
    seen.add(sig)
    t64 = -(-M // 64) * -(-Cout // 64)
    best = min(ts, key=ts.get)
    lines.append((t64, Cout, K, [ts[t] for t in tiles], best, src, "%s [M=%d %s]" % (name, M, geo)))
lines.sort(key=lambda r: (r[0], r[1], r[2]))
print("t64   Cout  K     " + "  ".join("t%-5d" % t for t in tiles) + " best  source layer")
for t64, Cout, K, ms, best, src, lay in lines:
    print("%5d %5d %5d  %s  t%-3d %s %s" % (t64, Cout, K, "  ".join("%.4f" % v for v in ms), best, src, lay))
