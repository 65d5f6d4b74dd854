"""Counter bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the HBM-bound stages bench.py times (roofline_hbm), per STEP:
   python tools/hbm_stage_traffic.py <fetch_dir> <write_dir> <forward steps of the profiled process> <out.json>
<forward steps> = EVERY forward the profiled process ran: tools/profile_round.sh runs bench.py with --warmup 2 --steps 10 and bench.py adds its
single-stream roofline pass of another 10 steps: 22.  (Round 4 passed 12: every figure it wrote is 22 / 12 = 1.83x too large -- the "1.6-1.9x
counter-over-algorithmic bytes" of RoIAlign and mask_logits_select in VERDICT r4 were this divisor, not over-fetch.)  A stage kernel that runs a fixed
number of times per forward must divide evenly: checked.
Stages that map to kernels of their own are summed by kernel name; stages built from shared kernels (top-k, NMS) are left out (null in the bench
line).  bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> B): the gfx950 wide-read correction is calibrated for 16 B per lane loads -- RoIAlign's gathers,
the mask / paste kernels' vector loads -- and an upper bound where a kernel reads narrower."""
import collections, csv, glob, json, os, sys
STAGES = {
    "roi_align 7x7 (box head)": [("roi_align_kernel", 0), ("roi_align_f16_c8_kernel", 0)],
    "roi_align 14x14 (mask head)": [("roi_align_kernel", 1), ("roi_align_f16_c8_kernel", 1)],
    "paste_masks (Masker: resize + threshold + paste, whole uint8 planes)": [("paste_masks_kernel", None)],
    "mask_logits_select (1x1 -> the label's channel + sigmoid)": [("mask_logits_select", None)],
    "yolact_masks (proto @ coeff -> sigmoid -> crop -> upsample -> threshold, whole uint8 planes)": [("yolact_proto_masks", None), ("yolact_upsample_masks", None)],
    "front end (uint8 -> resize / normalise / pad -> fp32 input)": [("preprocess_u8_kernel", None)],
}


def rows(d, counter):
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].split("::")[-1]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024.0))
    return {k: [v for _, v in sorted(vs)] for k, vs in out.items()}


def main():
    f, w, steps = rows(sys.argv[1], "FETCH_SIZE"), rows(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])
    out = {}
    for label, parts in STAGES.items():
        tot, found = 0.0, False
        for kname, which in parts:
            for k in f:
                if k.startswith(kname) or ("isegmi" in k and kname in k):   # (some kernels reach the CSV under their mangled names)
                    fv, wv = f[k], w.get(k, [0.0] * len(f[k]))
                    per = len(fv) // steps if steps else 0
                    if per == 0:
                        continue
                    if which is not None and len(fv) % steps != 0:
                        raise SystemExit("%s: %d launches do not divide over %d forward steps -- wrong step count?" % (k, len(fv), steps))
                    idx = range(len(fv)) if which is None else [i for i in range(len(fv)) if i % per == which]
                    tot += sum(2 * fv[i] + wv[i] for i in idx) / steps
                    found = True
        if found:
            out[label] = int(tot)
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
