"""Data-parallel equivalence of the HIP training step on the one GPU of the box: two FRESH child processes share GPU 0 and
all-reduce their flat gradients over gloo on device tensors; after one step they must hold what a single process holds after the
same step on the whole batch (examples/conformer/dataset.py:552-553 shards batch[rank::world]; train_one_step.py:38 reduces the
gradients; BatchNorm statistics stay per rank).  The 8-GPU RCCL run of the driver goes through the same BucketedAllReduce."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(world, tmp_path, mode="bfloat16"):
    port = _free_port()
    outs = [str(tmp_path / ("%s_w%d_r%d.pt" % (mode, world, r))) for r in range(world)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), str(world), str(port), outs[r], mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for p in procs:
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, log[-3000:]
    import torch

    return [torch.load(o) for o in outs]


def test_single_process_step_is_bit_reproducible(tmp_path):
    """Every parameter-gradient reduction of the step is per-workgroup partials + a fixed-order sum (no float atomics): two fresh
    processes produce bit-identical gradients, updates and BatchNorm statistics."""
    import torch

    (a,) = _run_world(1, tmp_path)
    os.rename(str(tmp_path / "bfloat16_w1_r0.pt"), str(tmp_path / "first.pt"))
    (b,) = _run_world(1, tmp_path)
    assert a["loss"] == b["loss"] and torch.equal(a["grad"], b["grad"]) and torch.equal(a["delta"], b["delta"])
    for x, y in zip(a["bn_mean"] + a["bn_var"], b["bn_mean"] + b["bn_var"]):
        assert torch.equal(x, y)


def test_two_ranks_equal_one_rank_float32_mode(tmp_path):
    """The same comparison as below in the float32 validation mode, where no bf16 rounding can amplify the different summation
    order of (2 ranks x half a batch) against (1 rank x the whole batch): gradients within 1e-4 relative, and NOT ONE element of
    Adam's first applied update (|update| = lr) changes sign outside the tensors whose true gradient is identically zero."""
    (one,) = _run_world(1, tmp_path, "float32")
    r0, r1 = _run_world(2, tmp_path, "float32")
    assert (r0["grad"] == r1["grad"]).all() and not one["overflow"] and not r0["overflow"]
    g2, g1 = r0["grad"].double(), one["grad"].double() * 2.0
    assert float((g2 - g1).norm() / g1.norm()) <= 1e-4
    import torch

    live = torch.ones(one["size"], dtype=torch.bool)
    for lo, hi in one["zero_spans"]:
        live[lo:hi] = False
    # ... and elements whose gradient is exactly zero in both runs (padding of the flat buffer, unused vocabulary rows)
    live &= (g1 != 0) | (g2 != 0)
    # a sign flip needs |g| below the float32 round-off of the reductions: compare where the gradient is above 1e-6 of the tensor scale
    strong = live & (g1.abs() > 1e-6 * float(g1.abs().max()))
    d2, d1 = r0["delta"].double(), one["delta"].double()
    assert int(((d2 - d1).abs() > 0.5 * r0["lr"])[strong].sum()) == 0
    assert float((d2 - d1)[strong].norm() / d1[strong].norm()) <= 1e-3


def test_two_ranks_equal_one_rank_on_the_whole_batch(tmp_path):
    (one,) = _run_world(1, tmp_path)
    r0, r1 = _run_world(2, tmp_path)
    # the reduced gradient is the same tensor on both ranks, and the two ranks saw different losses (different labels)
    assert (r0["grad"] == r1["grad"]).all() and r0["loss"] != r1["loss"]
    assert not one["overflow"] and not r0["overflow"] and not r1["overflow"]
    # loss: mean over the whole batch = mean of the ranks' means (equal shard sizes)
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - one["loss"]) <= 1e-4 * abs(one["loss"])
    # all-reduce(SUM) of per-rank means = world x the whole-batch mean gradient (same loss scale); 1/world is folded into Adam
    g2, g1 = r0["grad"].double(), one["grad"].double() * 2.0
    assert float((g2 - g1).norm() / g1.norm()) <= 2e-3
    # ... so the applied update is the single-process update.  This is Adam's first applied step, |update| = lr for every element
    # with a non-zero gradient: round-off can only flip the sign of elements whose gradient is noise (key / depthwise biases,
    # whose true gradient is zero) - measured 0.1 % of the elements
    d2, d1 = r0["delta"].double(), one["delta"].double()
    moved = d1.abs() > 0
    flipped = ((d2 - d1).abs() > 0.5 * r0["lr"]) & moved
    assert float(d1.abs().max()) > 0.9 * r0["lr"] and int(flipped.sum()) <= 5e-3 * int(moved.sum())
    assert float((d2 - d1)[~flipped].norm() / d1.norm()) <= 3e-2
    assert (r0["delta"] == r1["delta"]).all()  # replicas stay in lock-step
    # BatchNorm running statistics are per rank (no SyncBN in the reference): equal here because the shards' inputs are equal
    for a, b, c in zip(r0["bn_mean"], r1["bn_mean"], one["bn_mean"]):
        assert (a == b).all() and float((a - c).abs().max()) <= 1e-3 * float(c.abs().max() + 1e-6)
    # buckets: every element exactly once, launched in backward order (last block first, then head, then front end)
    order = r0["order"]
    pos = 0
    for lo, hi in sorted(order):
        assert lo == pos
        pos = hi
    assert pos == r0["size"]
    block_los = [lo for lo, _ in order[:2]]
    assert block_los == sorted(block_los, reverse=True) and order[-1][0] == 0


def test_two_ranks_hybrid_steps_with_changing_label_widths_tables_on_and_off(tmp_path):
    """The data-parallel hybrid step when the decoder is replayed at the recorded label width and walked at the others while the
    encoder's backward is replayed (round 6: the combination that computed a wrong ca_kv gradient before its fix): on each of two ranks
    the launch tables change no bit of twelve steps' losses and masters, every step reduces every element of the flat gradient
    exactly once on either path, and both ranks hold the same weights."""
    import torch

    on = _run_world(2, tmp_path, "hybrid_tables")
    for r in range(2):
        os.rename(str(tmp_path / ("hybrid_tables_w2_r%d.pt" % r)), str(tmp_path / ("on_r%d.pt" % r)))
    off = _run_world(2, tmp_path, "hybrid_walked")
    for a, b in zip(on, off):
        assert a["losses"] == b["losses"] and torch.equal(a["master"], b["master"])
        for res in (a, b):
            for spans in res["spans"]:
                pos = 0
                for lo, hi in spans:  # the reduced spans tile the flat gradient
                    assert lo == pos and hi > lo
                    pos = hi
                assert pos == res["size"]
    assert torch.equal(on[0]["master"], on[1]["master"])
