"""Per-stage MFMA-roofline table from a tools/conv_report.py listing (dev tool): python tools/stage_table.py <report.txt> [peak TF/s]"""
import re, sys
PEAK = float(sys.argv[2]) if len(sys.argv) > 2 else 157.3
STAGES = [("proto", r"^proto"), ("heads", r"^prediction|head_cat"), ("fpn", r"fpn"), ("rpn", r"^rpn"), ("box", r"roi_heads\.box"), ("mask", r"roi_heads\.mask"),
          ("stem", r"stem|^backbone\.conv1"), ("res2", r"backbone\.(layers\.0|body\.layer1)\."), ("res3", r"backbone\.(layers\.1|body\.layer2)\."),
          ("res4", r"backbone\.(layers\.2|body\.layer3)\."), ("res5", r"backbone\.(layers\.3|body\.layer4)\.")]
acc = {}
for l in open(sys.argv[1]):
    m = re.match(r"(\S+) \[.*\]\s+([\d.]+) GF\s+([\d.]+) ms", l)
    if not m: continue
    name, gf, ms = m.group(1), float(m.group(2)), float(m.group(3))
    st = next((s for s, pat in STAGES if re.search(pat, name)), "other")
    a = acc.setdefault(st, [0.0, 0.0, 0]); a[0] += gf; a[1] += ms; a[2] += 1
tg = sum(a[0] for a in acc.values()); tm = sum(a[1] for a in acc.values())
for st, (gf, ms, n) in acc.items():
    print("%-6s %3d convs %8.1f GF %7.3f ms %6.1f TF/s  frac %.3f" % (st, n, gf, ms, gf / ms, gf / ms / PEAK))
print("%-6s           %8.1f GF %7.3f ms %6.1f TF/s  frac %.3f" % ("total", tg, tm, tg / tm, tg / tm / PEAK))
