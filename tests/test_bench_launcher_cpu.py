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
    # the N > 1 fields of the train_dp object come out of bench.comm_metrics (run here over gloo on a stand-in engine): all present,
    # the all-reduce really ran (5 rounds of SUM over 2 ranks on a buffer of ones: 2^5 per element)
    tdp = line["train_dp_multi"]
    for key in ("ms_per_step_no_allreduce", "exposed_comm_ms", "allreduce"):
        assert key in tdp, tdp
    for key in ("ms", "bytes", "buckets", "bus_GBps", "xgmi_link_GBps", "frac_of_one_link"):
        assert key in tdp["allreduce"], tdp
    assert tdp["allreduce"]["buckets"] == 14 and tdp["allreduce"]["bytes"] == 14 * 1000 * 4 and tdp["allreduce"]["bus_GBps"] >= 0
    assert line["grad_sum"] == 14 * 1000 * 2.0 ** 5


def test_world_size_mismatch_is_an_error():
    res = _run(["--gpus", "4"], {"MA_BENCH_DRY": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


def test_single_gpu_does_not_spawn():
    res = _run([], {"MA_BENCH_DRY": "1"})
    assert res.returncode == 0 and json.loads(res.stdout.strip())["n_gpus"] == 1
