"""Child process of tests/test_dp_equivalence_gpu.py (not a test module): one data-parallel rank of the HIP training step.

  python tests/dp_worker.py <rank> <world> <port> <out.pt> [bfloat16 | float32]

All ranks share GPU 0 and talk over gloo on DEVICE tensors (the box has one GPU; the driver's 8-GPU run uses nccl = RCCL with
the same code path: BucketedAllReduce only sees torch.distributed).  Global batch = 4 utterances built so that each rank's shard
batch[rank::world] has the SAME inputs (so BatchNorm's per-rank batch statistics equal the whole batch's) but DIFFERENT labels
(so the ranks' gradients differ and the all-reduce matters)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def global_batch():
    import torch

    from oracle import conformer_oracle as C

    g = torch.Generator().manual_seed(21)
    tlen, vocab = 131, 97
    u = torch.randn(2, tlen, 80, generator=g)
    lens = [tlen, tlen - 30]
    for i, n in enumerate(lens):
        u[i, n:] = 0
    xs = torch.stack([u[0], u[0], u[1], u[1]])              # batch[0::2] = batch[1::2] = (u0, u1)
    mask = torch.zeros(4, 1, tlen)
    for i, n in enumerate([lens[0], lens[0], lens[1], lens[1]]):
        mask[i, 0, :n] = 1
    sub = C.subsample_mask(mask)
    ys_lens = torch.tensor([9, 7, 6, 5], dtype=torch.int32)
    ys = torch.full((4, 9), -1, dtype=torch.int32)
    for i, n in enumerate(ys_lens.tolist()):
        ys[i, :n] = torch.randint(1, vocab, (n,), generator=g, dtype=torch.int32)
    return xs, ys, sub, ys_lens


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    compute_type = sys.argv[5] if len(sys.argv) > 5 else "bfloat16"
    import torch

    torch.cuda.set_device(0)
    if world > 1:
        import torch.distributed as dist

        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mindaudio_amd.conformer.asr_model import create_asr_model, shard_batch
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    torch.manual_seed(5)  # same initial weights on every rank (examples/conformer/train.py:56)
    model = create_asr_model(80, 97, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=2)).cuda()
    eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=1, dropout_rate=0.0, positional_dropout_rate=0.0,
                                world_size=world, rank=rank, compute_type=compute_type)
    order = []
    launch = eng.reducer.launch
    eng.reducer.launch = lambda lo, hi: (order.append((lo, hi)), launch(lo, hi))[1]
    xs, ys, sub, ys_lens = (c.cuda() for c in shard_batch(global_batch(), rank, world))
    cols = (xs, ys, None, None, None, None, sub, None, None, ys_lens, None)
    before = eng.fp.master.clone()
    eng.step(*cols)                      # global_step 0: lr = 0 (scheduler_factory.py:44-50)
    order.clear()
    loss, cond, scale, overflow, lr = eng.step(*cols)
    # tensors whose true gradient is identically zero (a bias in front of a softmax over keys / in front of a BatchNorm): their
    # computed gradient is round-off noise of either sign
    zero_spans = []
    for li in range(eng.L):
        lo = eng.fp.index["l%d.qkv_b" % li][0]
        zero_spans.append((lo + eng.d, lo + 2 * eng.d))                       # linear_k.bias
        lo, _, n = eng.fp.index["l%d.dw_b" % li]
        zero_spans.append((lo, lo + n))                                       # depthwise_conv.bias
    torch.save({"loss": float(loss), "overflow": overflow, "scale": scale, "lr": lr, "order": order, "zero_spans": zero_spans,
                "grad": eng.fp.grad.cpu(), "delta": (eng.fp.master - before).cpu(), "size": eng.fp.size,
                "bn_mean": [m.cpu() for m in eng.bn_mean], "bn_var": [v.cpu() for v in eng.bn_var]}, out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
