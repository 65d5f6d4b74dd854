#!/bin/bash
# Where the FPN heads' RoIAlign launches spend their time (tools/roi_align_bench.py under rocprofv3 --pmc, one counter group per pass):
# wave cycles (active / issue stalls / parked at a wait), VALU and VMEM instruction counts, L1 (TCP) accesses against its requests to L2, L2 hits / misses.
root=$(pwd); out=$root/gpurun_out/roicnt; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args=${1:-2}
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/g$i -o run -- python3 $root/tools/roi_align_bench.py $args > $out/g$i.log 2>&1 || { echo "group $i failed: $grp"; tail -3 $out/g$i.log; continue; }
done
python3 - "$out" <<'PY'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "roi_" not in k: continue
        name = "tab" if "tab_kernel" in k else ("prep" if "prep" in k else "plain")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        # the table launches come in three groups (row order / random / roi_prep order): report the last third
        vv = v[len(v) * 2 // 3:] if name == "tab" else v
        print("   %-34s %14.0f  (avg of %d launches)" % (c, sum(vv) / len(vv), len(vv)))
PY
