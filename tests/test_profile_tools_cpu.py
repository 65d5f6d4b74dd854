"""The evidence pipeline's book-keeping (tools/hbm_stage_traffic.py): synthetic rocprofv3 counter CSVs -> per-stage bytes.  Round 4's file was told a
wrong step count and dealt the two RoIAlign launches of a step to the wrong heads; the tool now checks divisibility and deals launches by kernel."""
import csv
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "hbm_stage_traffic.py")
PLAIN = "void isegmi::roi_align_kernel<2>(isegmi::RoiLevels, float const*, int const*, int, int, int, int, int, int, int, int, int, float*, int*)"
TAB7 = "void isegmi::roi_align_tab_kernel<7, 7>(isegmi::RoiLevels, HIP_vector_type<int, 4u> const*, int const*, int const*, int, int, int, int, float*)"
TAB7H = "_ZN6isegmi24roi_align_f16_tab_kernelILi7ELi7EEEvNS_10RoiLevelsHEPK15HIP_vector_typeIiLj4EEPKiS7_iiiiPDF16_"
TAB14H = "_ZN6isegmi24roi_align_f16_tab_kernelILi14ELi14EEEvNS_10RoiLevelsHEPK15HIP_vector_typeIiLj4EEPKiS7_iiiiPDF16_"
PREP = "isegmi::roi_prep_kernel(float const*, int const*, int, isegmi::RoiPrepLevels, int, int, int, int, int, int, int*, HIP_vector_type<int, 4u>*)"
SEL = "isegmi::mask_logits_select_kernel(float const*, int, int, float const*, float const*, int const*, float*)"
FRONT = "isegmi::preprocess_u8_kernel(unsigned char const*, int, int, int, float*, int, int, int, int, long, isegmi::F3, isegmi::F3, int)"


def _write(d, counter, launches):
    """launches: [(kernel, KiB)] in dispatch order"""
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "run_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for i, (k, v) in enumerate(launches):
            w.writerow([i + 1, k, counter, v])


def _run(tmp_path, fetch, write, steps):
    _write(str(tmp_path / "f"), "FETCH_SIZE", fetch)
    _write(str(tmp_path / "w"), "WRITE_SIZE", write)
    out = str(tmp_path / "o.json")
    r = subprocess.run([sys.executable, TOOL, str(tmp_path / "f"), str(tmp_path / "w"), str(steps), out], capture_output=True, text=True)
    return r, (json.load(open(out)) if r.returncode == 0 else None)


def test_plain_launches_go_box_then_mask_and_the_front_end_counts_per_launch(tmp_path):
    steps = 4
    seq_f, seq_w = [], []
    for _ in range(steps):
        seq_f += [(PLAIN, 100), (SEL, 40), (PLAIN, 10)]          # KiB: box, select, mask
        seq_w += [(PLAIN, 50), (SEL, 1), (PLAIN, 20)]
    seq_f += [(FRONT, 7)] * 3; seq_w += [(FRONT, 26)] * 3           # a loop of its own: not a multiple of the forward steps
    r, o = _run(tmp_path, seq_f, seq_w, steps)
    assert r.returncode == 0, r.stderr + r.stdout
    assert o["roi_align 7x7 (box head)"] == (2 * 100 + 50) * 1024
    assert o["roi_align 14x14 (mask head)"] == (2 * 10 + 20) * 1024
    assert o["mask_logits_select (1x1 -> the label's channel + sigmoid)"] == (2 * 40 + 1) * 1024
    assert o["front end (uint8 -> resize / normalise / pad -> fp32 input)"] == (7 + 26) * 1024   # narrow loads: FETCH x 1


def test_table_driven_heads_take_their_prep_launch(tmp_path):
    steps = 3
    f, w = [], []
    for _ in range(steps):   # fp32 default: box head = prep + table launch, mask head = the plain launch
        f += [(PREP, 1), (TAB7, 48), (PLAIN, 10)]
        w += [(PREP, 2), (TAB7, 100), (PLAIN, 20)]
    r, o = _run(tmp_path, f, w, steps)
    assert r.returncode == 0, r.stderr + r.stdout
    assert o["roi_align 7x7 (box head)"] == (2 * 1 + 2 + 2 * 48 + 100) * 1024 and o["roi_align 14x14 (mask head)"] == (2 * 10 + 20) * 1024
    f, w = [], []
    for _ in range(steps):   # fp16 default: both heads from tables (mangled names, as rocprofv3 prints the _Float16 kernels); prep launches in head order
        f += [(PREP, 1), (TAB7H, 70), (PREP, 3), (TAB14H, 25)]
        w += [(PREP, 2), (TAB7H, 200), (PREP, 4), (TAB14H, 80)]
    r, o = _run(tmp_path, f, w, steps)
    assert r.returncode == 0, r.stderr + r.stdout
    assert o["roi_align 7x7 (box head)"] == (2 * 1 + 2 + 2 * 70 + 200) * 1024 and o["roi_align 14x14 (mask head)"] == (2 * 3 + 4 + 2 * 25 + 80) * 1024


def test_the_zero_fill_of_the_mask_planes_counts_with_the_stage_that_writes_them(tmp_path):
    steps = 2
    PASTE = "isegmi::paste_masks_kernel(float const*, float const*, int const*, int, int, int, int, int, float, unsigned char*, int*)"
    FILL = "__amd_rocclr_fillBufferAligned"
    f = [(FILL, 0), (FILL, 0), (PASTE, 3)] * steps
    w = [(FILL, 1000), (FILL, 1000), (PASTE, 30)] * steps
    r, o = _run(tmp_path, f, w, steps)
    assert r.returncode == 0, r.stderr + r.stdout
    assert o["paste_masks (Masker: resize + threshold + paste, whole uint8 planes)"] == (2 * 3 + 30 + 2000) * 1024


def test_a_wrong_step_count_is_refused(tmp_path):
    f = [(PLAIN, 100), (PLAIN, 10)] * 22
    w = [(PLAIN, 50), (PLAIN, 20)] * 22
    r, _ = _run(tmp_path, f, w, 12)     # round 4's mistake: 12 where the process ran 22 forwards
    assert r.returncode != 0 and "do not divide" in (r.stderr + r.stdout)
    r, o = _run(tmp_path, f, w, 22)
    assert r.returncode == 0 and o["roi_align 7x7 (box head)"] == 250 * 1024
