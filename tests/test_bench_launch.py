"""bench.py --gpus N must produce an N-rank run by itself: outside torchrun the parent starts N fresh child
processes of the file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and relays rank 0's one line; a world that differs from
--gpus is an error, never a silent one-GPU measurement.  `--launch-dry-run` exercises the launcher and the rendezvous it
sets up (gloo all-reduce over the ranks) without touching a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 4])
def test_gpus_n_starts_n_ranks_and_prints_one_line(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "3", "--warmup", "2", "--launch-dry-run"],
                       env=clean_env(), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["ranks"] == list(range(n))
    kids = d["children"]
    assert [k["local_rank"] for k in kids] == list(range(n))            # one GPU each
    assert all(k["world"] == n and k["sum_of_ranks_plus_1"] == n * (n + 1) // 2 for k in kids)      # the ranks met each other
    assert len({k["pid"] for k in kids}) == n and os.getpid() not in {k["pid"] for k in kids}      # fresh processes
    assert len({k["master"] for k in kids}) == 1 and kids[0]["master"].startswith("127.0.0.1:")


def test_world_that_differs_from_gpus_is_refused():
    env = clean_env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "3"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "WORLD_SIZE=2 but --gpus 1" in r.stderr
    env.update(WORLD_SIZE="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "3"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "WORLD_SIZE=1 but --gpus 8" in r.stderr


def test_a_failing_rank_fails_the_launch():
    """a child that dies takes the launch down with a non-zero exit (and the other ranks with it) instead of a partial line:
    here every child fails at once -- there is no GPU in the CPU suite, and without --launch-dry-run the ranks need one"""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "2", "--workers", "1", "--launch-timeout", "240"],
                       env=clean_env(), capture_output=True, text=True, timeout=280)
    if r.returncode == 0:
        pytest.skip("GPUs are visible here: the launch succeeded")
    assert r.stdout.strip() == "" and "rank exit codes" in r.stderr
