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


def hybrid_steps(rank, world, out, tables):
    """Twelve hybrid (0.3 CTC + attention) steps whose label width changes from step to step (10 / 7 / 13 columns): launch tables
    recorded and replayed for the encoder, the decoder replayed only at the recorded width and walked otherwise - with the gradient
    buckets of BOTH going through the all-reduce.  Saves the losses, the final masters and, per step, the spans that were reduced."""
    import torch

    from mindaudio_amd.conformer.asr_model import create_asr_model, shard_batch
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    torch.manual_seed(5)
    model = create_asr_model(80, 97, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=2), ctc_weight=0.3,
                             decoder_conf=dict(attention_heads=4, linear_units=512, num_blocks=2, dropout_rate=0.1,
                                               positional_dropout_rate=0.1), lsm_weight=0.1).cuda()
    eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=2, dropout_rate=0.1, positional_dropout_rate=0.1, world_size=world,
                                rank=rank, seed=3)
    eng.block_tables = tables
    order = []
    launch = eng.reducer.launch
    eng.reducer.launch = lambda lo, hi: (order.append((lo, hi)), launch(lo, hi))[1]
    xs, ys, sub, ys_lens = shard_batch(global_batch(), rank, world)
    losses, spans = [], []
    for k in range(12):
        lmax = (9, 9, 9, 6, 9, 12, 6, 9, 12, 12, 9, 6)[k]
        lens = torch.clamp(ys_lens, max=lmax)
        b, eos = ys.shape[0], 96
        ys_w = torch.full((b, lmax), -1, dtype=torch.int32)
        ys_in = torch.full((b, lmax + 1), eos, dtype=torch.int32)
        ys_out = torch.full((b, lmax + 1), -1, dtype=torch.int32)
        ys_masks = torch.zeros(b, 1, lmax + 1)
        for i, n in enumerate(lens.tolist()):
            ys_w[i, :n] = ys[i, :n]
            ys_in[i, 1:n + 1] = ys[i, :n]
            ys_out[i, :n] = ys[i, :n]
            ys_out[i, n] = eos
            ys_masks[i, 0, :n + 1] = 1
        ys_sub = (ys_masks.bool() & torch.tril(torch.ones(lmax + 1, lmax + 1, dtype=torch.bool))[None]).float()
        cols = (xs, ys_w, ys_in, ys_out, None, None, sub, ys_sub, ys_masks, lens, None)
        order.clear()
        losses.append(float(eng.step(*(c.cuda() if c is not None else None for c in cols))[0]))
        spans.append(sorted(order))
    torch.cuda.synchronize()
    torch.save({"losses": losses, "master": eng.fp.master.cpu(), "spans": spans, "size": eng.fp.size}, out)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    compute_type = sys.argv[5] if len(sys.argv) > 5 else "bfloat16"
    import torch

    torch.cuda.set_device(0)
    if world > 1:
        import torch.distributed as dist

        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if compute_type.startswith("hybrid"):
        hybrid_steps(rank, world, out, compute_type.endswith("tables"))
        if world > 1:
            dist.destroy_process_group()
        return
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
