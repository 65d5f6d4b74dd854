"""Host logic of the device-side COCO output (no GPU): the record block's layout (mirror of csrc/results.cpp), records -> COCO dicts against
the host-array path, the test loader's batch schedule, the torch-free rendezvous, and the oracle's RLE restatement against hand-computed
answers and isegmi/coco.py."""
import multiprocessing as mp
import os
import re

import numpy as np
import pytest


def test_oracle_rle_known_answers_and_agreement_with_host_encoder():
    from isegmi import coco
    from oracle import ora
    m = np.array([[0, 1], [1, 1]], np.uint8)                       # column-major 0 1 1 1
    assert list(ora.rle_encode(m)) == [1, 3] and ora.rle_to_string([1, 3]) == "13"
    assert list(ora.rle_encode(np.ones((2, 2), np.uint8))) == [0, 4]          # first pixel set: leading 0
    assert list(ora.rle_encode(np.zeros((3, 2), np.uint8))) == [6]
    assert ora.rle_to_string([32]) == "P1" and ora.rle_to_string([0, 4]) == "04"
    # the fourth count on is a difference to the count two back: 5 100 2000 40-100=-60 ...; negative values use the sign bit rule
    counts = [5, 100, 2000, 40, 1, 70000, 3]
    s = ora.rle_to_string(counts)
    assert coco.rle_from_string(s) == counts and s == coco.rle_to_string(counts)
    rng = np.random.default_rng(0)
    for h, w in ((1, 1), (3, 5), (64, 64), (65, 63), (138, 138)):
        for dens in (0.0, 0.05, 0.5, 1.0):
            mm = (rng.uniform(0, 1, (h, w)) < dens).astype(np.uint8)
            a = ora.rle_encode(mm)
            assert list(a) == coco.rle_counts(mm) and int(a.sum()) == h * w
            assert ora.rle_to_string(a) == coco.rle_to_string(a)
            assert np.array_equal(coco.rle_decode({"size": [h, w], "counts": ora.rle_to_string(a)}), mm)


@pytest.mark.parametrize("kind,mscore", [(2, False), (1, False), (1, True)])
def test_record_block_roundtrip_and_results(kind, mscore):
    from isegmi import coco
    from isegmi.dist import CocoRecordError, coco_record_layout, pack_coco_records, unpack_coco_records
    rng = np.random.default_rng(kind)
    n, K, h, w = 3, 5, 40, 50
    count = np.array([4, 0, 2], np.int32)
    masks = (rng.uniform(0, 1, (n, K, h, w)) < 0.4).astype(np.uint8)
    masks[:, :, 10:30, 5:45] = 1
    if kind == 2:
        box = np.sort(rng.uniform(0, 40, (n, K, 4)).astype(np.float32), -1)[..., [0, 1, 2, 3]]
        label = rng.integers(1, 81, (n, K)).astype(np.int32)
    else:
        box = np.sort(rng.integers(0, 40, (n, K, 4)), -1).astype(np.int64)
        label = rng.integers(0, 80, (n, K)).astype(np.int32)
    score = np.sort(rng.uniform(0.1, 1, (n, K)).astype(np.float32), -1)[:, ::-1]
    ms = rng.uniform(0, 1, (n, K)).astype(np.float32) if mscore else None
    chars, so = b"", [0]
    for i in range(n):
        for k in range(K):
            if k < count[i]:
                chars += coco.rle_to_string(coco.rle_counts(masks[i, k])).encode()
            so.append(len(chars))
    cap = len(chars) + 100
    status = np.array([0, len(chars), 0, 0], np.int32)
    buf = pack_coco_records(status, box, count, score, label, np.array(so, np.int32), chars, n, K, kind, ms, cap)
    secs, coff, total = coco_record_layout(n, K, kind, mscore, cap)
    assert buf.size == total and coff % 8 == 0 and all(off % 4 == 0 for off, _, _, _ in secs.values()) and secs["box"][0] % 8 == 0
    rec = unpack_coco_records(buf, n, K, kind, mscore, cap)
    assert rec["chars"] == chars and np.array_equal(rec["box"], box) and np.array_equal(rec["count"], count)
    ids, sizes = [7, 8, 9], [(h, w)] * n
    got = coco.results_from_records(rec, ids, sizes, kind, K)
    want = []
    for i in range(n):
        c = count[i]
        if kind == 2:
            want += coco.maskrcnn_results(ids[i], box[i, :c], score[i, :c], label[i, :c], masks[i, :c])
        else:
            want += coco.yolact_results(ids[i], label[i, :c], score[i, :c], box[i, :c], masks[i, :c], None if ms is None else ms[i, :c])
    assert got == want and len(got) == 6
    if kind == 1:   # eval.py --score_threshold / --top_k on the records
        thr = float(np.sort(score[0, :4])[1])
        f = coco.results_from_records(rec, ids, sizes, kind, K, score_threshold=thr, top_k=1)
        assert [d["image_id"] for d in f] == [7, 9] and all(d["score"] > thr for d in f)
    assert coco.results_from_records(rec, [7, None, 9], sizes, kind, K) == [d for d in want if d["image_id"] != 8]
    bad = buf.copy(); bad[8:12] = np.array([2], np.int32).view(np.uint8)
    with pytest.raises(CocoRecordError):
        unpack_coco_records(bad, n, K, kind, mscore, cap)
    # an all-zero block (a rank without a batch in the last step) is a valid, empty record
    z = unpack_coco_records(np.zeros(total, np.uint8), n, K, kind, mscore, cap)
    assert coco.results_from_records(z, ids, sizes, kind, K) == []


def test_schedule_batches_is_the_grouped_batch_sampler():
    from isegmi.pipeline import schedule_batches
    assert schedule_batches(["a", "b", "a", "a", "b", "a"], 2) == [[0, 2], [1, 4], [3, 5]]
    assert schedule_batches([0] * 5, 2) == [[0, 1], [2, 3], [4]]
    assert schedule_batches([], 4) == []
    b = schedule_batches([i % 3 for i in range(20)], 4)
    assert sorted(i for x in b for i in x) == list(range(20)) and all(len({i % 3 for i in x}) == 1 for x in b) and [x[0] for x in b] == sorted(x[0] for x in b)


def _rdzv_rank(rank, world, port, q, calls=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ISEGMI_RUN_ID="t")
    from isegmi.dist import rendezvous_unique_id
    for c in range(calls):
        uid = rendezvous_unique_id(rank, world, lambda c=c: bytes([c]) + bytes(range(1, 128)), timeout=30.0)
        q.put((rank, c, uid))


def _free_port_base():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_rendezvous_without_torch():
    """rank 0's 128-byte id reaches the other ranks over the loopback TCP hand-off, whichever side starts first, and TWICE in a row (a job makes
    one rendezvous per communicator: the call number in the hello keeps a fast rank's second call from being answered by rank 0's first)"""
    port = _free_port_base()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rdzv_rank, args=(r, 3, port, q, 2)) for r in (1, 2, 0)]
    for p in ps[:2]:
        p.start()
    import time
    time.sleep(0.3)
    ps[2].start()
    got = [q.get(timeout=60) for _ in range(6)]
    for p in ps:
        p.join(30)
    assert sorted((r, c) for r, c, _ in got) == [(r, c) for r in range(3) for c in range(2)]
    assert all(uid == bytes([c]) + bytes(range(1, 128)) for _, c, uid in got)


def _stale_listener(port, q):
    """what a crashed earlier run could leave behind for a while: a live listener on the first rendezvous port that answers a DIFFERENT job's hello"""
    import socket
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    srv.bind(("127.0.0.1", port + 1))
    srv.listen(8)
    srv.settimeout(20)
    q.put("up")
    try:
        while True:
            c, _ = srv.accept()
            with c:
                msg = c.recv(256)
                if b" stale-run " in msg:      # only ranks of ITS run get its id
                    c.sendall(b"\xee" * 128)
    except OSError:
        pass


def test_rendezvous_ignores_a_fresh_stale_peer():
    """the file rendezvous accepted a < 60 s old left-over of a crashed run.  The TCP hand-off holds no state: a listener of
    another (crashed, still dying) run on the first rendezvous port is asked, does not know this run's hello, and is passed over -- rank 0 of
    THIS run binds the next port and its id is the one every rank gets."""
    port = _free_port_base()
    ctx = mp.get_context("spawn")
    q0 = ctx.Queue()
    stale = ctx.Process(target=_stale_listener, args=(port, q0))
    stale.start()
    assert q0.get(timeout=30) == "up"
    q = ctx.Queue()
    ps = [ctx.Process(target=_rdzv_rank, args=(r, 2, port, q)) for r in (1, 0)]
    for p in ps:
        p.start()
    got = [q.get(timeout=60) for _ in range(2)]
    for p in ps:
        p.join(30)
    stale.terminate()
    stale.join(10)
    assert all(uid == bytes([0]) + bytes(range(1, 128)) for _, _, uid in got)


def test_rendezvous_times_out_without_rank0():
    from isegmi import dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port_base()), ISEGMI_RUN_ID="nobody")
    with pytest.raises(TimeoutError):
        dist.rendezvous_unique_id(1, 2, lambda: b"\0" * 128, timeout=1.0)


def test_package_imports_no_torch_and_no_oracle():
    """north_star: host code is Python over a thin C ABI -- no PyTorch, no Triton in the product; the oracle is test infrastructure"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "instancesegmentation-jittor_amd", "isegmi")
    pat = re.compile(r"^\\s*(import|from)\\s+(torch|triton|oracle)\\b", re.M)
    for f in sorted(os.listdir(root)):
        if f.endswith(".py"):
            src = open(os.path.join(root, f)).read()
            assert not pat.search(src), f


def test_launcher_sets_rank_environment_and_propagates_failure(tmp_path):
    """python -m isegmi.launch --nproc N ...: N fresh processes with the RANK / WORLD_SIZE environment, the TCP rendezvous works between
    them, a failing rank takes the group down and its code is the launcher's; --init-timeout kills a group whose ranks never report a
    communicator (what a hung ncclCommInitRank looks like from outside) with code 125"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "instancesegmentation-jittor_amd")
    w = tmp_path / "w.py"
    w.write_text("import os, sys\nsys.path.insert(0, %r)\nfrom isegmi.dist import rendezvous_unique_id, notify_launcher\n"
                 "r, n = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\nuid = rendezvous_unique_id(r, n, lambda: bytes([7]) * 128, timeout=30)\n"
                 "open(os.path.join(%r, 'ok%%d' %% r), 'wb').write(uid)\n"
                 "mode = sys.argv[1] if len(sys.argv) > 1 else ''\n"
                 "if mode == 'hang':\n    import time; time.sleep(60)\n"
                 "notify_launcher('comm %%d' %% r)\nimport time; time.sleep(0.3)\n"
                 "sys.exit(3 if (r == 1 and mode == 'fail') else 0)\n" % (pkg, str(tmp_path)))
    env = dict(os.environ, PYTHONPATH=pkg)
    r = subprocess.run([sys.executable, "-m", "isegmi.launch", "--nproc", "3", "--init-timeout", "60", str(w)], env=env, timeout=120)
    assert r.returncode == 0 and all((tmp_path / ("ok%d" % i)).read_bytes() == bytes([7]) * 128 for i in range(3))
    r = subprocess.run([sys.executable, "-m", "isegmi.launch", "--nproc", "2", str(w), "fail"], env=env, timeout=120)
    assert r.returncode == 3
    r = subprocess.run([sys.executable, "-m", "isegmi.launch", "--nproc", "2", "--init-timeout", "3", str(w), "hang"], env=env, timeout=120)
    assert r.returncode == 125
