"""bench.py's launch contract (no GPU needed: both checks happen before torch or the library is touched)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=120)


def test_world_size_mismatch_is_an_error():
    # under a launcher with another world size the run would measure something else than it reports
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)


def test_single_gpu_run_without_gpu_fails_loudly():
    # no CPU fallback: on a box without an MI355X the bench refuses instead of timing something else
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    r = _run(["--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0 and "no CPU path" in (r.stderr + r.stdout)
