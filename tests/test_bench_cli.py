"""bench.py's launcher contract, the part that needs no GPU: a --gpus / WORLD_SIZE mismatch is refused before anything touches a device."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_must_match_the_launchers_world_size():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "4"})
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    r = _run(["--gpus", "0"], {})
    assert r.returncode != 0


def test_without_a_gpu_the_bench_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
