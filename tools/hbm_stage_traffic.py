"""Counter bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the HBM-bound stages bench.py times (roofline_hbm), per STEP:
   python tools/hbm_stage_traffic.py <fetch_dir> <write_dir> <forward steps of the profiled process> <out.json>
<forward steps> = EVERY forward the profiled process ran: tools/profile_round.sh runs bench.py with --warmup 2 --steps 10 and bench.py adds its
single-stream roofline pass of another 10 steps: 22.  (Round 4 passed 12; with 12 the RoIAlign launches of a step were also dealt to the wrong head,
profiles/r05_experiments.txt 7 and 10.)  A stage kernel that runs a fixed number of times per forward must divide evenly: checked.
Stages that map to kernels of their own are summed by kernel name; stages built from shared kernels (top-k, NMS) are left out (null in the bench
line).  bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> B): the gfx950 wide-read correction is calibrated for 16 B per lane loads -- RoIAlign's gathers,
the mask / paste kernels' vector loads -- and an upper bound where a kernel reads narrower.
The zero fill of the uint8 mask planes (hipMemsetAsync -> __amd_rocclr_fillBufferAligned; the only memsets of a forward) belongs to the stage that then
writes the masks into them: paste_masks (Mask R-CNN) / yolact_masks.
RoIAlign: a head is either one plain launch (roi_align_kernel<2> / roi_align_f16_c8_kernel: box head first, mask head second when both are plain)
or roi_prep_kernel + the table-driven launch (<7, 7> box head, <14, 14> mask head); roi_prep_kernel's launches of a step go to the heads in order."""
import collections, csv, glob, json, os, re, sys

BOX, MASK = "roi_align 7x7 (box head)", "roi_align 14x14 (mask head)"
SIMPLE = {
    "paste_masks (Masker: resize + threshold + paste, whole uint8 planes)": r"paste_masks_kernel",
    "mask_logits_select (1x1 -> the label's channel + sigmoid)": r"mask_logits_select",
    "yolact_masks (proto @ coeff -> sigmoid -> crop -> upsample -> threshold, whole uint8 planes)": r"yolact_proto_masks|yolact_upsample_masks",
}
# one launch per batch, timed by bench.py in a loop of its own: per LAUNCH.  Its loads are single bytes (the uint8 taps), not 16 B per lane: the x2 correction
# does not apply -- calibrated on its known input, raw FETCH_SIZE 7.5 MB against the 6.4 MB uint8 batch of Mask R-CNN bs = 2 (taps shared between rows)
FRONT = "front end (uint8 -> resize / normalise / pad -> fp32 input)"


def rows(d, counter):
    path = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]) * 1024.0))
    return {k: [v for _, v in sorted(vs)] for k, vs in out.items()}


def main():
    f, w, steps = rows(sys.argv[1], "FETCH_SIZE"), rows(sys.argv[2], "WRITE_SIZE"), int(sys.argv[3])

    def launches(pat, fetch_factor=2):
        """[bytes of launch 0, launch 1, ...] over every kernel whose name matches, in dispatch order per kernel"""
        out = []
        for k in f:
            if re.search(pat, k):
                wv = w.get(k, [0.0] * len(f[k]))
                out.append([fetch_factor * a + b for a, b in zip(f[k], wv)])
        return out

    def per_step(vals, which=None, of=None):
        if len(vals) % steps:
            raise SystemExit("%d launches do not divide over %d forward steps -- wrong step count?" % (len(vals), steps))
        per = len(vals) // steps
        if of is not None and per != of:
            raise SystemExit("expected %d launches per step, found %d" % (of, per))
        return sum(v for i, v in enumerate(vals) if which is None or i % per == which) / steps

    out = {}
    for label, pat in SIMPLE.items():
        ls = launches(pat)
        if ls:
            out[label] = int(sum(per_step(v) for v in ls))
            if "uint8 planes" in label:
                out[label] += int(sum(sum(v) for v in launches(r"fillBufferAligned")) / steps)
    ls = launches(r"preprocess_u8_kernel", 1)
    if ls:
        out[FRONT] = int(sum(sum(v) / len(v) for v in ls))
    box = sum((launches(p) for p in (r"roi_align_tab_kernel<7, 7>", r"roi_align_f16_tab_kernel(<7, 7>|ILi7ELi7E)")), [])
    mask = sum((launches(p) for p in (r"roi_align_tab_kernel<14, 14>", r"roi_align_f16_tab_kernel(<14, 14>|ILi14ELi14E)")), [])
    plain = launches(r"roi_align_kernel<2>|roi_align_kernelILi2E|roi_align_f16_c8_kernel")
    prep = launches(r"roi_prep_kernel")
    tot = {BOX: sum(per_step(v) for v in box), MASK: sum(per_step(v) for v in mask)}
    for v in plain:
        if box and mask:
            raise SystemExit("plain RoIAlign launches next to both table-driven heads")
        if box or mask:
            tot[MASK if box else BOX] += per_step(v, of=1)
        else:
            tot[BOX] += per_step(v, 0, of=2); tot[MASK] += per_step(v, 1, of=2)
    for v in prep:
        heads = [h for h, t in ((BOX, box), (MASK, mask)) if t]
        for i, h in enumerate(heads):
            tot[h] += per_step(v, i, of=len(heads))
    if box or mask or plain:
        out[BOX], out[MASK] = int(tot[BOX]), int(tot[MASK])
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
