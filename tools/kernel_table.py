"""Per-kernel table from one rocprofv3 --stats run plus the FETCH_SIZE / WRITE_SIZE PMC passes of the same command:
average duration, HBM-side bytes per launch and the implied GB/s.  Note the three runs are separate processes
(durations from the un-countered --stats run); conv FETCH is x2-corrected (wide loads), other kernels are reported
raw AND x2 because their access widths are uncalibrated (MI355X_MICROARCH.md, HBM section).

usage: python tools/kernel_table.py <stats_dir> <fetch_dir> <write_dir> <out.md>
"""
import collections
import csv
import glob
import os
import sys


def one(d, pat):
    return glob.glob(os.path.join(d, "**", pat), recursive=True)[0]


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            d[k][0] += 1
            d[k][1] += float(r["Counter_Value"])
    return d


def main():
    stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(one(sys.argv[1], "*kernel_stats.csv")))}
    f = agg(one(sys.argv[2], "*counter_collection.csv"), "FETCH_SIZE")
    w = agg(one(sys.argv[3], "*counter_collection.csv"), "WRITE_SIZE")
    rows = []
    for k, r in stats.items():
        avg_us = float(r["AverageNs"]) / 1e3
        fb = f[k][1] / max(f[k][0], 1) * 1024 if k in f else 0.0
        wb = w[k][1] / max(w[k][0], 1) * 1024 if k in w else 0.0
        rows.append((float(r["TotalDurationNs"]), k, int(r["Calls"]), avg_us, fb, wb))
    rows.sort(reverse=True)
    with open(sys.argv[4], "w") as o:
        o.write("| kernel | calls | avg us | % time | FETCH MB/launch (raw) | WRITE MB/launch | GB/s (raw fetch+write) | GB/s (2x fetch+write) |\n|---|---|---|---|---|---|---|---|\n")
        tot = sum(r[0] for r in rows)
        for t, k, c, us, fb, wb in rows:
            if t / tot < 0.0005:
                continue
            o.write("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.0f | %.0f |\n" % (k[:60], c, us, 100 * t / tot, fb / 1e6, wb / 1e6, (fb + wb) / us / 1e3, (2 * fb + wb) / us / 1e3))
    print(open(sys.argv[4]).read())


if __name__ == "__main__":
    main()
