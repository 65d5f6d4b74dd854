"""Stream -> hardware-queue placement.  The ROCm runtime folds a process's streams onto four in-order hardware queues by what
the process created before; round 3 measured 980 / 922 / 979 / 920 / 908 img/s (Yolact bs 8) with 0 .. 4 foreign streams created before the engine.
The engine now builds its stream set from probed candidates (csrc/engine.cpp, acquire_streams): the layout -- and with it the throughput -- must
not depend on the process's history."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["main", "side0", "side1", "side2", "tail", "heads", "hs0", "hs1", "hs2", "copy"]


def test_engine_stream_layout_is_the_designed_partition(ffi):
    """main's hardware queue carries neither the tail, nor the heads, nor the copy stream, nor side0 (the backbone's projection shortcuts); the tail's
    queue carries neither heads nor copy -- as probed on the live engine (isegmi_engine_stream_layout), with foreign streams created first."""
    hip = C.CDLL("libamdhip64.so")
    foreign = []
    for _ in range(3):
        st = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(st)) == 0
        foreign.append(st)
    from isegmi.weights import yolact_state_dict
    from isegmi.yolact import Yolact
    net = Yolact(yolact_state_dict(1234), max_batch=1, input_size=200)
    q = (C.c_int32 * 10)()
    ffi.check(ffi.lib().isegmi_engine_stream_layout(net._h, q, 10))
    cls = dict(zip(NAMES, q))
    net.close()
    for st in foreign:
        hip.hipStreamDestroy(st)
    if len(set(q)) < 4:
        pytest.skip("fewer than four hardware queues on this box: %s" % cls)
    assert cls["main"] not in (cls["tail"], cls["heads"], cls["copy"]), cls
    assert cls["tail"] not in (cls["heads"], cls["copy"]), cls
    assert cls["side0"] not in (cls["tail"], cls["heads"], cls["copy"]), cls


def test_layout_does_not_depend_on_foreign_streams(ffi):
    """A fresh process per k = 0, 3 foreign streams created before the engine (tools/stream_layout_probe.py): the probed partition is the designed
    one either way.  The throughput of the k runs is printed, not asserted (these kernels sit at the package power cap and the clock
    state moves a run by several per cent with no code change; round 4 measured a 0.3 % spread over k = 0 .. 4, round 3 6-10 % -- the comparison
    lives in the tool and in profiles/, the invariant that produced it is what is tested)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stream_layout_probe.py"), "0,3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = re.findall(r"k=(\d+)\s+([\d.]+) img/s\s+queues: (.*)", r.stdout)
    assert len(rows) == 2, r.stdout + r.stderr
    for k, val, qs in rows:
        cls = {n: int(c) for n, c in (t.split(":") for t in qs.split())}
        if len(set(cls.values())) < 4:
            pytest.skip("fewer than four hardware queues on this box: %s" % cls)
        assert cls["main"] not in (cls["tail"], cls["heads"], cls["copy"], cls["side0"]), (k, cls)
        assert cls["tail"] not in (cls["heads"], cls["copy"]), (k, cls)
    print("img/s by foreign streams:", {k: v for k, v, _ in rows})
