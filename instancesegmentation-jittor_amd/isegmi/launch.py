"""One process per GPU without torch: `python -m isegmi.launch --nproc 8 -m isegmi.cli test_net --config-file ... --images ...`

Starts N child processes (fresh interpreters; nothing in this parent touches HIP) with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT / ISEGMI_RUN_ID in their environment -- the variables `python -m torch.distributed.run` would set -- waits for them, kills the
whole group if one fails or the time limit passes, and exits with the first non-zero code.  The ranks meet through isegmi.dist's TCP
rendezvous (MASTER_PORT + 1 ...) and exchange their detection records over RCCL (the GPUs of one node).
`--init-timeout S`: a watchdog for the one call that can block for ever -- ncclCommInitRank with a rank missing or with mismatched ids: every
rank reports its first communicator to this parent (a datagram, isegmi.dist.notify_launcher); if not all of them have within S seconds, the
whole group is killed and the exit code is 125.  The children are fresh processes; nothing that touched the GPU is ever re-executed."""
import argparse
import os
import signal
import socket
import subprocess
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="isegmi.launch")
    ap.add_argument("--nproc", type=int, required=True, help="ranks = GPUs of this node")
    ap.add_argument("--timeout", type=float, default=0.0, help="seconds before the ranks are killed (0: none)")
    ap.add_argument("--init-timeout", type=float, default=0.0, help="seconds within which every rank must have created its first RCCL communicator (0: no watchdog)")
    ap.add_argument("-m", dest="module", default=None, help="run a module (python -m) instead of a script")
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    cmd = [sys.executable] + (["-m", a.module] if a.module else []) + a.rest
    if not a.module and not a.rest:
        ap.error("nothing to launch")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    run_id = "%d_%d" % (os.getpid(), int(time.time()))
    note = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
    note.bind(("127.0.0.1", 0))
    note.setblocking(False)
    procs = []
    for r in range(a.nproc):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ISEGMI_RUN_ID=run_id, ISEGMI_LAUNCH_NOTIFY=str(note.getsockname()[1]), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd, env=env, start_new_session=True))
    t0, code = time.time(), 0
    ready = set()
    try:
        while any(p.poll() is None for p in procs):
            bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
            if bad or (a.timeout > 0 and time.time() - t0 > a.timeout):
                code = bad[0] if bad else 124
                break
            try:
                while True:
                    m = note.recv(64).decode(errors="replace").split()
                    if len(m) == 2 and m[0] == "comm":
                        ready.add(m[1])
            except (BlockingIOError, OSError):
                pass
            if a.init_timeout > 0 and len(ready) < a.nproc and time.time() - t0 > a.init_timeout:
                sys.stderr.write("isegmi.launch: %d of %d ranks have an RCCL communicator after %.0f s; killing the group\n" % (len(ready), a.nproc, a.init_timeout))
                code = 125
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # a rank that died before the collective leaves the others blocked in RCCL: take the rest down with it
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
            p.wait()
        note.close()
    code = code or next((p.returncode for p in procs if p.returncode), 0)
    raise SystemExit(code)


if __name__ == "__main__":
    main()
