"""`python bench.py --gpus N` started plainly starts its own ranks (a child `torch.distributed.run`).  Without a
GPU the rehearsal gets as far as the point where a rank needs one and ends there with a clear message and a
non-zero exit code; nothing is measured here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_starts_two_ranks_and_relays_their_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("rehearsal of the no-GPU ending; on a GPU box the real run is the driver's")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ACMPC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--lines-per-gpu", "1000", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "as a child process" in r.stderr
    assert "rank 0 of 2: no GPU visible" in r.stderr and "rank 1 of 2: no GPU visible" in r.stderr
    assert "{" not in r.stdout                      # no JSON line from a run that measured nothing


def test_a_launcher_with_another_rank_count_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 2 but the launcher started 3" in r.stderr
