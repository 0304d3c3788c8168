#!/usr/bin/env python3
"""cfg 4 of SURVEY §8(d): data-parallel CTC training step of Conformer-small on synthetic AISHELL-shaped batches.

  python tools/train_bench.py [--steps K --warmup W --batch 40 --frames 1024 --vocab 4233 --blocks 12]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py ...

One rank per GPU; every rank steps on its own (batch, frames, 80) shard (bucket 1024 of conformer.yaml), gradients are
all-reduced over RCCL in per-block buckets overlapped with the backward pass.  Prints one JSON line on rank 0."""
import argparse, json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=40)
    ap.add_argument("--frames", type=int, default=1024)
    ap.add_argument("--vocab", type=int, default=4233)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--ctc-weight", type=float, default=1.0, help="< 1: hybrid loss with a 6-block TransformerDecoder")
    args = ap.parse_args()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    torch.manual_seed(777)  # same initial weights on every rank (train.py:56)
    hybrid = args.ctc_weight != 1.0
    model = create_asr_model(80, args.vocab, dict(output_size=256, attention_heads=4, linear_units=2048,
                                                  num_blocks=args.blocks), ctc_weight=args.ctc_weight,
                             decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6,
                                               dropout_rate=args.dropout, positional_dropout_rate=args.dropout)
                             if hybrid else None, lsm_weight=0.1 if hybrid else 0.0).to(dev)
    eng = ConformerCTCTrainStep(model, dropout_rate=args.dropout, positional_dropout_rate=args.dropout,
                                world_size=world, rank=rank)
    rng = np.random.RandomState(1234 + rank)
    b, t = args.batch, args.frames
    xs = torch.from_numpy(rng.randn(b, t, 80).astype(np.float32)).to(dev)
    lens = rng.randint(int(0.7 * t), t + 1, b)
    lens[0] = t
    t2 = ((t - 3) // 2 + 1 - 3) // 2 + 1
    masks = torch.zeros(b, 1, t2)
    for i, n in enumerate(lens):
        masks[i, 0, :(n - 1) // 4] = 1  # frames 4j < n (dataset.py:625)
        xs[i, n:] = 0
    masks = masks.to(dev)
    ylens = rng.randint(5, 31, b).astype(np.int32)
    ys = np.full((b, 30), -1, np.int32)
    for i, n in enumerate(ylens):
        ys[i, :n] = rng.randint(1, args.vocab - 1, n)
    sos = eos = args.vocab - 1
    ys_in = np.full((b, 31), eos, np.int32)
    ys_out = np.full((b, 31), -1, np.int32)
    ys_m = np.zeros((b, 1, 31), np.float32)
    for i, n in enumerate(ylens):
        ys_in[i, 0] = sos
        ys_in[i, 1:n + 1] = ys[i, :n]
        ys_out[i, :n] = ys[i, :n]
        ys_out[i, n] = eos
        ys_m[i, 0, :n + 1] = 1
    ys_sub = (ys_m.astype(bool) & np.tril(np.ones((31, 31), bool))[None]).astype(np.float32)
    cols = (xs, torch.from_numpy(ys).to(dev), torch.from_numpy(ys_in).to(dev), torch.from_numpy(ys_out).to(dev), None, None,
            masks, torch.from_numpy(ys_sub).to(dev), torch.from_numpy(ys_m).to(dev), torch.from_numpy(ylens).to(dev), None)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        losses.append(float(eng.step(*cols)[0]))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = eng.step(*cols)
    barrier()
    dt = time.perf_counter() - t0
    # host time of a step's launches alone: forward_backward() enqueues everything and never synchronises
    barrier()
    t1 = time.perf_counter()
    for _ in range(3):
        eng.forward_backward(cols[0], cols[1], cols[6], cols[9], None, grad_scale=1024.0, ys_in_pad=cols[2], ys_out_pad=cols[3],
                             ys_sub_masks=cols[7], ys_masks=cols[8])
    host_ms = (time.perf_counter() - t1) / 3 * 1e3
    barrier()
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        nparam = eng.fp.size
        print(json.dumps({
            "metric": "utterances/s, Conformer-small CTC training step (fwd + bwd + all-reduce + Adam)",
            "value": round(world * b * args.steps / dt, 1), "unit": "utterances/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "host_enqueue_ms_fwd_bwd": round(host_ms, 3),
            "higher_is_better": True,
            "scaling": "weak", "dtype": "bf16 matmuls, f32 master/grads/optimizer", "data": "synthetic",
            "config": {"workload": "bucket-1024 batch (%d, %d, 80) per rank, V=%d, %d blocks, dropout %.2f, %s, "
                                   "Adam + ASRWarmupLR + dynamic loss scale" % (b, t, args.vocab, args.blocks, args.dropout, "hybrid CTC %.1f / attention (6-block decoder, label smoothing 0.1)" % args.ctc_weight if hybrid else "pure CTC"),
                       "global_batch": b * world, "flat_params": nparam,
                       "grad_bytes_allreduced_per_step": nparam * 4 if world > 1 else 0},
            "first_losses": [round(v, 3) for v in losses[:3]], "last_loss": round(float(out[0]), 3),
            "loss_scale": out[2], "overflow": out[3]}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
