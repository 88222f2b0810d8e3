"""bench.py --gpus N without torchrun: the GPU-free parent spawns N ranks with
the torch.distributed environment, rank 0 owns stdout, a failing rank's exit
code comes back, and a --gpus / WORLD_SIZE mismatch is refused (VERDICT round 1
item 2; reference semantics: main_2d.py:89-94,147-149)."""
import json
import os
import subprocess
import sys

from tests.conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(RPDE_BENCH_DRYRUN="1", **env)
    return subprocess.run([sys.executable, BENCH, *args], env=e, capture_output=True, text=True, timeout=300)


def test_gpus_n_spawns_n_ranks_with_the_distributed_environment():
    r = _run(["--gpus", "3", "--steps", "2"])
    assert r.returncode == 0, r.stderr
    recs = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda d: d["rank"])
    assert [d["rank"] for d in recs] == [0, 1, 2]
    assert all(d["world"] == 3 and d["local_rank"] == d["rank"] for d in recs)
    assert len({d["master"] for d in recs}) == 1 and recs[0]["master"].startswith("127.0.0.1:")


def test_a_failing_rank_fails_the_launch_with_its_exit_code():
    r = _run(["--gpus", "2"], RPDE_BENCH_FAIL_RANK="1", RPDE_BENCH_FAIL_CODE="7")
    assert r.returncode == 7


def test_single_rank_needs_no_spawn_and_mismatch_is_refused():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0 and json.loads(r.stdout.strip())["world"] == 1
    r = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_spawn_ranks_helper_propagates_and_terminates(tmp_path):
    from rpde.launch import rank_env, spawn_ranks
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK'])\n"
                      "assert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if r == 0:\n    sys.exit(5)\n"
                      "time.sleep(60)\n")          # rank 1 would hang: the launcher must end it
    assert spawn_ranks(str(script), [], 2) == 5
    env = rank_env(1, 4, 1234, base={})
    assert env["RANK"] == "1" and env["WORLD_SIZE"] == "4" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_host_thread_pool_is_sized_by_the_usable_cores(monkeypatch):
    """rpde.launch.host_cores / limit_host_threads: the affinity mask capped by the cgroup quota and RPDE_CPU_THREADS,
    shared between the ranks of a node; the pool is only ever made smaller"""
    import torch
    from rpde import launch
    before = torch.get_num_threads()
    try:
        monkeypatch.setenv("RPDE_CPU_THREADS", "6")
        n = launch.host_cores()
        assert 1 <= n <= 6
        torch.set_num_threads(max(before, 8))
        got = launch.limit_host_threads(world=4)
        assert got == max(1, n // 4) and torch.get_num_threads() == got
        torch.set_num_threads(1)
        assert launch.limit_host_threads(world=1) == 1          # never raised
    finally:
        torch.set_num_threads(before)
