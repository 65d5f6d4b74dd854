"""Static instruction mix of the conv kernels per basic block (dev tool):
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude -I<csrc> <file.hip> -o /tmp/k.s
   python tools/isa_count.py /tmp/k.s [name-substring]
Vector-ALU instructions between MFMAs take matrix-pipe issue slots (tools/microbench/mfma_switch.hip), so the per-block VALU
count next to the MFMA count is the number that matters."""
import re, sys
src = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = re.split(r'^(_Z\w+):.*$', src, flags=re.M)
for name, body in zip(funcs[1::2], funcs[2::2]):
    if want not in name:
        continue
    body = body.split(".Lfunc_end")[0]
    blocks, cur = [], ["entry", []]
    for l in body.split("\n"):
        t = l.strip()
        if not t or t.startswith((";", "//")):
            continue
        if re.match(r"^\.LBB\w+:", t):
            blocks.append(cur); cur = [t.split(":")[0], []]
        elif not t.startswith("."):
            cur[1].append(t.split()[0])
    blocks.append(cur)
    print(name)
    tot = [0, 0, 0, 0, 0]
    for label, ins in blocks:
        v = sum(i.startswith("v_") and "mfma" not in i for i in ins)
        m = sum("mfma" in i for i in ins)
        s = sum(i.startswith("s_") for i in ins)
        d = sum(i.startswith("ds_") for i in ins)
        b = sum(i.startswith(("buffer_", "global_", "flat_", "scratch_")) for i in ins)
        for k, x in enumerate((v, m, s, d, b)): tot[k] += x
        if len(ins) >= 8:
            print("  %-14s n %4d  valu %4d  mfma %3d  salu %4d  lds %3d  vmem %3d" % (label, len(ins), v, m, s, d, b))
    print("  %-14s         valu %4d  mfma %3d  salu %4d  lds %3d  vmem %3d" % ("TOTAL", *tot))
