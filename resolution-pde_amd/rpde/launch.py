"""One process per GPU without torchrun: start N copies of a script with the
torch.distributed environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), wait
for them and hand back the worst exit code.

The parent never touches the GPU (no HIP call, no ``torch.cuda`` query), so the
children are ordinary fresh processes; rank 0 inherits stdout and prints the
result line.  Counterpart of the reference's single-process
``nn.DataParallel`` set-up (main_2d.py:89-94,147-149), done the
one-process-per-GPU way.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import List, Optional, Sequence


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    return env


def spawn_ranks(script: str, argv: Sequence[str], world: int, poll_s: float = 0.2) -> int:
    """run ``python script *argv`` as ranks 0..world-1; returns 0 when every rank
    exited 0, else the first non-zero exit code seen (the other ranks are then
    terminated by PID -- a rank that died would leave them waiting in a collective)"""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, script, *argv], env=rank_env(r, world, port)))
    rc = 0
    alive = set(range(world))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for o in sorted(alive):
                    procs[o].terminate()
        if alive:
            time.sleep(poll_s)
    if rc != 0:
        deadline = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def host_cores() -> int:
    """cores this process may really use: affinity mask, capped by the cgroup CPU quota (and RPDE_CPU_THREADS)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    return max(1, min(n, int(os.environ.get("RPDE_CPU_THREADS", "64"))))


def limit_host_threads(world: int = 1) -> int:
    """torch sizes its intra-op pool by the machine's logical CPUs (256 on an MI355X host) even when the process may use
    16 of them (a container's CPU quota): every multi-threaded host op -- stacking a batch, a copy into a pinned buffer --
    then wakes hundreds of threads on a handful of cores and the launching thread loses 50-100 ms at a time, during which
    the GPU runs dry (profiles/mres_host_probe.py: mixed-resolution epochs of 237-277 ms against 173 ms).  One process
    per GPU: each gets its share of the usable cores.  Returns the thread count set."""
    import torch
    n = max(1, host_cores() // max(1, int(world)))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


def freeze_setup_garbage() -> None:
    """call once the model, the optimizer and the loaders exist: a full (generation-2) collection of a process that has
    imported torch walks ~10^6 objects -- 80-160 ms on these hosts (profiles/mres_host_probe.py), during which nothing is
    launched; a 3 ms step's queue runs dry.  Everything alive now is long-lived: collect once, then move it to the
    permanent generation so that later collections only look at what the steps create."""
    import gc
    gc.collect()
    gc.freeze()
