"""`python bench.py --gpus N` starts its own ranks (one process per GPU) before anything touches the GPU; a WORLD_SIZE that
disagrees with --gpus is an error, not a silent single-GPU run.  Checked here without a GPU: MA_BENCH_DRY=1 makes the ranks
rendezvous over gloo, all-reduce one number and print the line's `n_gpus`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=240):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout, text=True)


def test_gpus_2_spawns_two_ranks():
    res = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"MA_BENCH_DRY": "1"})
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_sum"] == 3.0  # ranks 0 and 1 both joined the all-reduce


def test_world_size_mismatch_is_an_error():
    res = _run(["--gpus", "4"], {"MA_BENCH_DRY": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


def test_single_gpu_does_not_spawn():
    res = _run([], {"MA_BENCH_DRY": "1"})
    assert res.returncode == 0 and json.loads(res.stdout.strip())["n_gpus"] == 1
