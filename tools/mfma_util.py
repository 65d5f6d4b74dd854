"""MfmaUtil per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)  (rocprofv3's own MfmaUtil expression).
usage: python tools/mfma_util.py <pmc_dir> <out.txt>"""
import collections, csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    d = per[r["Dispatch_Id"]]
    d[r["Counter_Name"]] = float(r["Counter_Value"]); d["k"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for v in per.values():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0:
        a = agg[v["k"]]; a[0] += 1; a[1] += v["SQ_VALU_MFMA_BUSY_CYCLES"]; a[2] += v["GRBM_GUI_ACTIVE"]; a[3] += v["dur"]
        a[4] = max(a[4], v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024))
with open(sys.argv[2], "w") as o:
    o.write("kernel | launches | MfmaUtil (time-weighted) | best single dispatch | clock GHz (GUI_ACTIVE/8/duration)\n")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][3]):
        if a[1] == 0:
            continue
        o.write("%s | %d | %.1f %% | %.1f %% | %.2f\n" % (k, a[0], 100 * a[1] / (a[2] / 8 * 1024), 100 * a[4], a[2] / 8 / a[3]))
print(open(sys.argv[2]).read())
