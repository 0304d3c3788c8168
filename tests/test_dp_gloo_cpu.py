"""CPU, world_size 2, gloo: the data-parallel pieces of the path that involve ranks — the strided batch shard of the
collate function (dataset.py:552-553) and the evaluation-loss all-reduce of create_asr_eval_net
(asr_model.py:355-371).  The network is a stand-in returning a per-rank loss (the HIP encoder needs a GPU)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mindaudio_amd.conformer.asr_model import ASREvalNet, shard_batch

    full = [torch.arange(12).reshape(6, 2), torch.arange(6)]
    mine = shard_batch(full, rank, world)

    class Net(torch.nn.Module):
        def forward(self, xs, ys):
            return (xs.float().sum() + ys.float().sum(), None)

    loss = ASREvalNet(Net(), world)(*mine)
    q.put((rank, [m.tolist() for m in mine], float(loss)))
    dist.destroy_process_group()


def test_shard_and_eval_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full_x, full_y = torch.arange(12).reshape(6, 2), torch.arange(6)
    assert res[0][1][0] == full_x[0::2].tolist() and res[1][1][0] == full_x[1::2].tolist()
    assert res[0][1][1] == full_y[0::2].tolist() and res[1][1][1] == full_y[1::2].tolist()
    want = (float(full_x.sum()) + float(full_y.sum())) / world  # all-reduce(SUM) / device_num
    assert res[0][2] == pytest.approx(want) and res[1][2] == pytest.approx(want)


def test_eval_net_single_rank_is_identity():
    from mindaudio_amd.conformer.asr_model import ASREvalNet

    class Net(torch.nn.Module):
        def forward(self, x):
            return x.sum()

    assert float(ASREvalNet(Net(), 1)(torch.ones(3))) == 3.0


# ---- gradient buckets of the training step (mindaudio_amd/train/engine.py) on gloo, world_size 2 ---------------------
def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mindaudio_amd.train.engine import BucketedAllReduce, FlatParams, bucket_spans, conformer_ctc_entries

    L = 3
    fp = FlatParams(conformer_ctc_entries(256, 512, L, 15, 4, 19, 101), "cpu")
    g = torch.Generator().manual_seed(100 + rank)
    fp.grad.copy_(torch.randn(fp.size, generator=g))
    mine = fp.grad.clone()
    red = BucketedAllReduce(fp.grad, world)
    for lo, hi in bucket_spans(fp, L):  # the order the backward pass launches them
        red.launch(lo, hi)
    covered = red.wait()
    q.put((rank, mine.numpy(), fp.grad.numpy().copy(), covered, fp.size))  # by value: the child exits right after
    dist.destroy_process_group()


def test_gradient_buckets_cover_everything_and_sum_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = torch.from_numpy(res[0][1]) + torch.from_numpy(res[1][1])
    for rank, _, reduced, covered, size in res:
        assert torch.allclose(torch.from_numpy(reduced), total)  # every element was reduced exactly once
        pos = 0
        for lo, hi in covered:  # the buckets tile the flat buffer: no gap, no overlap
            assert lo == pos
            pos = hi
        assert pos == size


def test_lr_schedule_and_loss_scale_rules():
    from mindaudio_amd.train.engine import DynamicLossScale, asr_warmup_lr

    assert asr_warmup_lr(0) == 0.0  # scheduler_factory.py:44-50 at global_step 0
    assert asr_warmup_lr(25000) == pytest.approx(1e-3)
    assert asr_warmup_lr(100) == pytest.approx(1e-3 * 25000 ** 0.5 * 100 * 25000 ** -1.5)
    assert asr_warmup_lr(100000) == pytest.approx(1e-3 * 25000 ** 0.5 * 100000 ** -0.5)
    s = DynamicLossScale(1024, 2, 3)
    s.update(True)
    assert s.scale == 512
    for _ in range(3):
        s.update(False)
    assert s.scale == 1024
    for _ in range(12):
        s.update(True)
    assert s.scale == 1.0  # floor (DynamicLossScaleUpdateCell: max(scale / factor, 1))
