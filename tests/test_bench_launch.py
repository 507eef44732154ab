"""bench.py's launch contract, without a GPU: `python bench.py --gpus N` must BE N ranks (round 5's script parsed --gpus and never
read it: started the way the driver starts it, an 8-GPU run would have measured one GPU and printed n_gpus 1).  --dry-launch takes
the same path up to the first HIP call -- self-launch under torch.distributed.run, rendezvous (gloo here), the barriers of a timed
region, one JSON line from rank 0 -- and leaves the kernels out."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def _line(out):
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"one JSON line expected, got {len(lines)}:\n{out.stdout[-2000:]}\n{out.stderr[-2000:]}"
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_becomes_n_ranks(n):
    out = _run(["--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-launch"])
    assert out.returncode == 0, out.stderr[-3000:]
    line = _line(out)
    assert line["n_gpus"] == n and line["gpus_asked_for"] == n and line["dry_launch"] is True
    assert line["steps"] == 3 and line["warmup"] == 1 and line["config"]["backend"] == "gloo"


def test_gpus_1_stays_one_process():
    out = _run(["--gpus", "1", "--dry-launch"])
    assert out.returncode == 0, out.stderr[-3000:]
    line = _line(out)
    assert line["n_gpus"] == 1 and line["config"]["backend"] == "none"  # (no launcher, no process group: the N = 1 path is what it was)


def test_the_drivers_own_launcher_still_works():
    """The contract's N > 1 form: torch.distributed.run around bench.py -- no second launch from inside."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--dry-launch"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert _line(out)["n_gpus"] == 2


def test_a_launcher_with_another_rank_count_is_refused():
    env = {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29534"}
    out = _run(["--gpus", "4", "--dry-launch"], env)
    assert out.returncode != 0 and "--gpus 4" in out.stderr and "1 rank" in out.stderr


def test_more_gpus_than_the_node_has_is_refused():
    """Without --dry-launch the parent counts the node's GPUs before it starts anything (none here)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("a node with GPUs")
    out = _run(["--gpus", "2", "--steps", "1"])
    assert out.returncode != 0 and "--gpus 2" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]  # no line at all rather than a one-GPU line
