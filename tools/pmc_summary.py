"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into profiles/<name>.json.

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
Units/corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports exactly half the bytes of wide (16 B/lane) coalesced reads -> doubled here for the conv
kernels, whose A/B tile loads are all dwordx4 (their 4-B residual loads are uncalibrated: upper bound).
"""
import collections
import csv
import json
import sys


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0]
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def main():
    f = agg(sys.argv[1], "FETCH_SIZE")
    w = agg(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes per launch (averaged over all launches of the kernel in the profiled run)",
           "correction": "FETCH_SIZE x2 (gfx950 wide-read undercount) applied to fetch_bytes_corrected", "kernels": {}}
    cf = cw = n = 0
    for k in f:
        nl = f[k][0]
        fb = f[k][1] / nl * 1024.0
        wb = w[k][1] / max(w[k][0], 1) * 1024.0
        out["kernels"][k] = {"launches": nl, "fetch_bytes_raw": round(fb), "fetch_bytes_corrected": round(2 * fb), "write_bytes": round(wb)}
        if any(t in k for t in ("conv_mfma", "conv_hybrid_kernel", "conv_group_kernel", "conv_f16_glds_kernel", "conv_f16_persist_kernel", "conv3x3_f16_strip_kernel", "bottleneck_f16_kernel", "stem_pool_f16_kernel")):  # every conv launch bench.py times
            cf += f[k][1] * 1024.0
            cw += w[k][1] * 1024.0
            n += nl
    out["conv_mfma_kernel_all"] = {"launches": n, "fetch_bytes_corrected": round(2 * cf / n), "write_bytes": round(cw / n),
                                   "hbm_bytes_per_launch": round((2 * cf + cw) / n)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["conv_mfma_kernel_all"]))


if __name__ == "__main__":
    main()
