"""Single-node launcher: one process per GPU (stdlib only -- nothing here may touch the GPU or import torch,
the children are started before any of that happens in the parent).

`spawn_ranks(n, argv)` starts `n` copies of `argv` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT set the way `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` would, relays rank 0's
stdout to this process's stdout (every rank's stderr goes to stderr), waits for all of them and returns 0 -- or, as
soon as one rank fails, terminates the others and returns that rank's code.  `bench.py --gpus N` uses it when it was not started by a launcher itself."""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv, env=None, stdout=None):
    if n < 1:
        raise ValueError("need at least one rank")
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(free_port()))
    base["WORLD_SIZE"] = str(n)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes on this driver)
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = sys.stdout if stdout is None else stdout
    # Rank 0's stdout is relayed by a reader thread while this thread watches EVERY child: when one rank ends with a
    # non-zero code (no device, a failed setup) the others would sit in their next collective until the backend's
    # time-out -- they are terminated and that code is returned.  (Fresh child processes; nothing is re-exec'd.)
    import threading
    import time

    def relay():
        for line in procs[0].stdout:                            # rank 0 prints the one JSON line
            out.write(line.decode() if isinstance(line, bytes) else line)
            out.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code)
                for q in live:
                    q.terminate()
        if live:
            time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    th.join(timeout=10)
    return rc
