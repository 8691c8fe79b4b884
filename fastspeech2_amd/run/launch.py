"""One process per GPU without an external launcher.

`python bench.py --gpus N` (no torchrun around it) has to become N ranks by itself: the process that was started
spawns N children of the same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set and
never touches the GPU itself (a parent that had initialised HIP would hold a context on GPU 0 for nothing).  Under
torchrun (WORLD_SIZE already set) nothing happens.  The reference's counterpart is single-process
nn.DataParallel(model, device_ids=range(ngpu)) (run/train_decode.py:257-262).
"""
import os
import socket
import subprocess
import sys
import time


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def bind_rank_cpus():
    """Gives this rank its share of the CPUs it is allowed to run on: the allowed set (os.sched_getaffinity, i.e. the
    container's or the job's cpuset) is cut into LOCAL_WORLD_SIZE contiguous parts and rank LOCAL_RANK keeps part
    LOCAL_RANK, so that eight ranks issuing launches from Python do not migrate over each other's cores (contiguous CPU
    numbers share a socket / NUMA node on the usual enumerations).  Call before torch is imported: threads created later
    inherit the mask.  A no-op for a single rank, without sched_setaffinity, or when a rank would get no CPU.
    Returns the list of CPUs kept (None when nothing was changed)."""
    try:
        n = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        r = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
        if n <= 1 or not hasattr(os, "sched_setaffinity") or os.environ.get("FS2_NO_CPU_BIND"):
            return None
        cpus = sorted(os.sched_getaffinity(0))
        per = len(cpus) // n
        if per < 1 or not 0 <= r < n:
            return None
        mine = cpus[r * per:(r + 1) * per]
        os.sched_setaffinity(0, mine)
        return mine
    except (OSError, ValueError):
        return None


def maybe_self_launch(n_ranks: int, argv=None) -> None:
    """Returns in a process that is (or does not need to be) a rank; otherwise runs the ranks and exits with 0 only if
    every rank did (a rank that fails takes the others down instead of leaving them in a barrier)."""
    if n_ranks <= 1 or "WORLD_SIZE" in os.environ:
        return
    argv = list(sys.argv if argv is None else argv)
    port = _free_port()
    procs = []
    for rank in range(n_ranks):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable] + argv, env=env))
    codes = [None] * n_ranks
    failed = False
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
                if codes[i] not in (None, 0):
                    failed = True
        if failed:
            for i, p in enumerate(procs):          # exactly the PIDs started here
                if codes[i] is None:
                    p.terminate()
            deadline = time.time() + 10
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[i] = p.wait()
            break
        time.sleep(0.05)
    bad = [(i, c) for i, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("self-launch: ranks failed (rank, exit code): %s\n" % bad)
        sys.exit(1)
    sys.exit(0)
