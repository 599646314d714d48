"""`python bench.py --gpus N` as the driver's SCALE tier gives it: without a launcher around it the script must start
its own ranks (one process per GPU, torch.distributed.run on 127.0.0.1) before making any GPU call, relay rank 0's
single JSON line and the children's exit code.  On CPU the hot path cannot run (no fallback), so the launcher, the
rendezvous and the gradient all-reduce are covered with --plumbing-only over gloo; the real two-rank step is the gpu test."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra, timeout=300):
    env = dict(os.environ, PSVO_DIST_BACKEND="gloo", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]      # (gloo prints connection chatter on stdout)
    return p, lines


@pytest.mark.timeout(400)
def test_bench_gpus2_launches_its_own_ranks_cpu():
    p, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--plumbing-only"], {})
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, lines                      # exactly one JSON line on stdout: rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["allreduce_ok"] is True


@pytest.mark.timeout(400)
def test_bench_relays_a_failing_rank_cpu():
    p, _ = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "C*"], {})
    # no --plumbing-only: on a CPU-only host every rank refuses (no CPU fallback); on a GPU host this is a real run
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_gpus2_real_step_on_one_card():
    """two ranks sharing the one card of the test box (gloo for the collective: RCCL refuses two ranks on one device):
    the whole bench path -- sharded batch, flat-gradient all-reduce, Adam, max-over-ranks timing -- prints n_gpus = 2"""
    p, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "2", "--workload", "tiny", "--no-cpu-baseline"], {})
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["value"] > 0
