#!/bin/bash
# FETCH_SIZE of the box head's RoIAlign forms (tools/roi_align_bench.py under rocprofv3 --pmc): per launch, x2 wide-read correction, MB
root=$(pwd); out=$root/gpurun_out/roifetch; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for a in "2" "2 clustered" "8 f16" "8 f16 clustered"; do
  i=$((i+1))
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/run$i -o run -- python3 $root/tools/roi_align_bench.py $a > $out/run$i.log 2>&1 || exit 1
  python3 - "$out/run$i" "$a" <<'PY'
import collections, csv, glob, os, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] == "FETCH_SIZE" and "roi_align" in r["Kernel_Name"]:
        rows["sliced" if "tab" in r["Kernel_Name"] else "plain"].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024 * 2 / 1e6))
print(sys.argv[2])
p = [v for _, v in sorted(rows["plain"])]
print("  plain            %.0f MB" % (sum(p) / len(p)))
s = [v for _, v in sorted(rows["sliced"])]
n = len(s) // 3
for j, name in enumerate(("row order", "random", "roi_prep")):
    print("  table, %-9s %.0f MB" % (name, sum(s[j * n:(j + 1) * n]) / n))
PY
done
