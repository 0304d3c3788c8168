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
