"""`python bench.py --gpus N` without a launcher must start the N ranks itself (fresh child processes through
torch.distributed.run, before anything touches a GPU) and relay rank 0's JSON line.  On this side of the GPU box the
ranks only meet (--check-launch, gloo): the data path needs an MI355X (tests/test_gpu_rccl.py runs the real thing)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_bench_starts_its_own_ranks():
    out = _run(["--gpus", "2", "--check-launch"])
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 alone reports
    rec = json.loads(lines[0])["launch_check"]
    assert rec["world"] == 2 and rec["env_world"] == 2 and rec["rank_sum"] == 1.0
    assert "starting 2 ranks" in out.stderr


def test_bench_refuses_a_world_that_contradicts_gpus():
    out = _run(["--gpus", "2", "--check-launch"], WORLD_SIZE="3", RANK="0")
    assert out.returncode != 0 and "WORLD_SIZE=3" in (out.stderr + out.stdout)


def test_single_rank_needs_no_launcher():
    out = _run(["--check-launch"])
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["launch_check"]["world"] == 1
