"""RCCL on ONE GPU: the data-parallel training step with its 14 asynchronous gradient all-reduces issued through
ProcessGroupNCCL (= RCCL on ROCm) at world size 1, as ReduceOp.AVG: an in-place SUM over one rank is short-circuited without any
device work, AVG (a pre-multiplied sum) makes RCCL launch its one-rank reduction kernel on its own stream - x * 1/1, the identity.
So the trained masters must be bit-identical to a run that issues no collective at all - which they are only if (i) RCCL's kernels
wait for the backward kernels the library launched on torch's current stream and (ii) the optimizer waits for the collectives
(mindaudio_amd/train/engine.py BucketedAllReduce; the reference's grad_reducer, mindaudio/utils/train_one_step.py:36-41,
examples/conformer/train.py:73-80).  Each run is a fresh child process started with `python -m torch.distributed.run`
(never a re-exec of a process that has touched the GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _train_line(extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--train", "--steps", "3",
           "--warmup", "1", "--no-cpu-baseline", "--train-digest"] + extra
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.startswith("{")]
    assert lines, res.stdout.decode()[-2000:]
    return json.loads(lines[-1])


def test_bucketed_allreduce_through_rccl_at_world_1_leaves_the_step_bit_identical():
    with_cc = _train_line(["--force-collective"])
    without = _train_line([])
    assert with_cc["n_gpus"] == 1 and with_cc["train_dp"]["force_collective"] is True
    assert "force_collective" not in without["train_dp"]
    assert with_cc["train_dp"]["last_loss"] == without["train_dp"]["last_loss"]
    assert with_cc["train_dp"]["masters_sha16"] == without["train_dp"]["masters_sha16"], (with_cc["train_dp"], without["train_dp"])
    # the all-reduce-alone leg (bus bandwidth, exposed communication) runs too: 14 buckets, 138 MB; at world 1 the ring factor is 0
    ar = with_cc["train_dp"]["allreduce"]
    assert ar["buckets"] == 14 and ar["bytes"] == with_cc["train_dp"]["grad_bytes"] and ar["ms"] > 0 and ar["bus_GBps"] == 0.0
    assert "exposed_comm_ms" in with_cc["train_dp"]
    # the roofline object of the training step is in the line
    r = with_cc["train_dp"]["roofline"]
    assert r["bound"] == "mfma" and 2.5e12 < r["algorithmic_flops_per_step"] < 3.2e12 and 0 < r["frac"] < 1


def test_rccl_device_kernels_really_run_beside_the_backward_pass(tmp_path):
    """the world-1 test must launch REAL RCCL kernels.  The same command under `rocprofv3 --kernel-trace`: every step
    issues 14 bucket all-reduces, each of which must appear as an RCCL device kernel in the trace, and some of them must have run
    concurrently with one of the library's backward kernels on another hardware queue (the condition N > 1 training creates)."""
    import shutil

    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        pytest.skip("rocprofv3 not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import rccl_trace_check

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # bench.py as the profiled program itself (a rank of a world of one: no launcher, no re-exec under the profiler)
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), TMPDIR=str(tmp_path))
    steps, warmup = 3, 1
    cmd = [prof, "--kernel-trace", "--output-format", "csv", "-d", str(tmp_path / "trace"), "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--train", "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline",
           "--force-collective"]
    res = subprocess.run(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    info = rccl_trace_check.analyse(str(tmp_path / "trace"))
    print(json.dumps(info))
    assert info["library_kernels"] > 1000
    # 14 buckets per step in the timed and warm-up steps (the no-all-reduce leg and the all-reduce-alone leg add none / SUM no-ops)
    assert info["rccl_kernels"] >= 14 * (steps + warmup), info
    # (how many of them overlapped one of the library's kernels in time is reported by tools/rccl_trace.sh -> profiles/; the one-rank
    # reduction of a 10 MB bucket takes a few microseconds, so the count depends on where the buckets fall)
