"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV of tools/e2e_host_time.py: where no convolution runs, and what runs then.
   python tools/e2e_timeline.py <kernel_trace.csv> [marker kernel substring = pad_c3_c4]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "pad_c3_"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("isegmi::", "").replace("void ", ""), r.get("Queue_Id", "")) for r in rows))
starts = [i for i, e in enumerate(ev) if mark in e[2]]
def is_rle(i0, i1): return any("rle_pack" in ev[i][2] for i in range(i0, i1))
for label, want in (("plain", False), ("e2e", True)):
    cands = [(starts[j], starts[j + 1]) for j in range(len(starts) - 1) if is_rle(starts[j], starts[j + 1]) == want]
    if len(cands) < 6: continue
    per = [ev[b][0] - ev[a][0] for a, b in cands[3:-2]]
    print("%s: %d steps, marker-to-marker interval mean %.3f ms" % (label, len(per), sum(per) / len(per) / 1e6))
    a, b = cands[len(cands) // 2]
    t0 = ev[a][0]
    conv = sorted((s, e) for s, e, n, q in ev[a:b] if ("conv_" in n or "conv3x3" in n) and "maskiou" not in n)
    busy_end, gaps = conv[0][0], []
    for s, e in conv:
        if s > busy_end + 15000: gaps.append((busy_end, s))
        busy_end = max(busy_end, e)
    print("  conv idle gaps > 15 us inside the step (%d), total %.3f ms:" % (len(gaps), sum(g[1] - g[0] for g in gaps) / 1e6))
    for g0, g1 in gaps:
        inside = [(s, e, n, q) for s, e, n, q in ev if s < g1 and e > g0 and "conv_" not in n and "conv3x3" not in n]
        print("   gap %.3f -> %.3f ms (%.0f us): %s" % ((g0 - t0) / 1e6, (g1 - t0) / 1e6, (g1 - g0) / 1e3, ", ".join("%s[q%s %.0fus]" % (n[:28], q, (e - s) / 1e3) for s, e, n, q in inside[:8])))
    print("  non-conv kernels of the step (start ms, dur us, queue):")
    for s, e, n, q in ev[a:b]:
        if "conv_" not in n and "conv3x3" not in n: print("   %7.3f %7.1f q%s %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, n[:60]))
    if want:
        print("  full listing around the step boundary (start ms, dur us, queue, kernel):")
        t1 = ev[b][0]
        for s, e, n, q in ev:
            if t1 - 1.2e6 < s < t1 + 0.9e6: print("   %8.3f %7.1f q%s %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, n[:70]))
