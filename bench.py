#!/usr/bin/env python3
"""bench.py — headline benchmark of the mindaudio hot path on MI355X.

Metric (BASELINE.json): utterances/s (16 kHz x 10 s), fbanks + Conformer forward.

One "step" = one pass of the hot path over one synthetic batch resident in HBM:
  waves (64, 160000) f32 --features.fbank(n_fft=512, hop=160, n_mels=80)--> (64, 80, 1001) dB
  --> (64, 1000, 80) --> Conformer-small encoder forward (12 blocks, d=256, 4 heads, ff=2048, k=15; eval mode,
  bf16 MFMA matmuls with float32 accumulation) --> (64, 249, 256).
N GPUs = N independent shards of 64 utterances (weak scaling; inference forward has no data-path collective and
the reference's batch-global top_db floor is per call, i.e. per rank — SURVEY §8e).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH, SAMPLES, N_FFT, HOP, N_MELS, SR = 64, 160000, 512, 160, 80, 16000
FRAMES = 1000
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16


def synth_batch(seed):
    return (0.1 * np.random.RandomState(seed).randn(BATCH, SAMPLES)).astype(np.float32)


def cpu_baseline(budget_s=20.0):
    """The oracle (CPU restatement of the reference algorithms) on a bounded sample of the same workload:
    fbank in the reference's own cost structure (float64 framing loop + per-column rFFT, 1 process) followed by
    the PyTorch-CPU float32 eager Conformer forward on all host threads."""
    import torch

    from oracle import conformer_oracle as C
    from oracle import speech_features as O

    n_utt = 4
    x = synth_batch(1234)[:n_utt]
    torch.manual_seed(0)
    enc = C.ConformerEncoder(80, 256, 4, 2048, 12).eval()
    mask = C.subsample_mask(torch.ones(n_utt, 1, FRAMES))
    done, total = 0, 0.0
    while total < budget_s and done < 4 * n_utt:
        t0 = time.perf_counter()
        feats = O.fbank_ref_cost(x, n_mels=N_MELS, n_fft=N_FFT, sample_rate=SR, hop_length=HOP)
        xs = torch.from_numpy(np.ascontiguousarray(feats.transpose(0, 2, 1)[:, :FRAMES]).astype(np.float32))
        with torch.no_grad():
            enc(xs, mask)
        total += time.perf_counter() - t0
        done += n_utt
    return {"value": round(done / total, 3), "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d utterances (10 s @16 kHz): oracle fbank_ref_cost (NumPy float64, 1 process) + oracle "
                      "Conformer-small forward (PyTorch-CPU float32 eager, %d threads), batches of %d"
                      % (done, torch.get_num_threads(), n_utt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import mindaudio_amd as ma
    from mindaudio_amd import _host, _lib, ops
    from mindaudio_amd.models import ConformerEncoder

    lib = _lib.load()
    torch.manual_seed(777)  # examples/conformer/train.py:56
    enc = ConformerEncoder(80, 256, 4, 2048, 12).eval().to(dev).prepare()
    x = torch.from_numpy(synth_batch(1234 + rank)).to(dev)
    t2 = ((FRAMES - 3) // 2 + 1 - 3) // 2 + 1
    masks = torch.ones(BATCH, 1, t2, device=dev)
    kw = dict(n_mels=N_MELS, n_fft=N_FFT, hop_length=HOP)

    def step():
        feats = ma.fbank(x, **kw)                                   # (64, 80, 1001) dB
        xs = feats.transpose(1, 2)[:, :FRAMES]                      # (64, 1000, 80) VIEW of the (64, 80, 1001) fbank output
        return enc(xs, masks)[0]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert tuple(out.shape) == (BATCH, t2, 256)

    def event_time(fn, reps):
        for _ in range(5):
            fn()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / reps * 1e-3  # seconds per launch

    # ---- roofline of the dominant kernel: ffn_packed_kernel (MFMA bound; ~45 % of the step), timed in the form the encoder
    #      launches it 11 times per forward: the last FFN of a block + the macaron FFN of the next + the three LayerNorms
    #      around them + linear_q/k/v of the next attention, one launch (the other 2 launches per forward are single FFNs).
    #      Algorithmic FLOPs per launch = 2 FFNs x 2 M (256*2048 + 2048*256) + 2 M 256*768, M = 64*249 rows. ---------------
    m, hid, nqkv = BATCH * t2, 2048, 768
    gen = torch.Generator(device=dev).manual_seed(99)
    rnd = lambda *shape: torch.randn(*shape, device=dev, generator=gen)
    wa1, wb1 = (rnd(hid, 256) / 16).bfloat16(), (rnd(hid, 256) / 16).bfloat16()
    wa2, wb2 = (rnd(256, hid) / 45).bfloat16(), (rnd(256, hid) / 45).bfloat16()
    wq, bq = (rnd(nqkv, 256) / 16).bfloat16(), rnd(nqkv)
    b1, b2 = rnd(hid), rnd(256)
    xres = rnd(m, 256)
    pa, pb, pq = ops.ffn_pack_weights(wa1, wa2), ops.ffn_pack_weights(wb1, wb2), ops.ffn_qkv_pack(wq)
    ln = (torch.ones(256, device=dev), torch.zeros(256, device=dev))
    gemm_s = event_time(lambda: ops.ffn_packed_pair(pa, b1, b2, pb, b1, b2, xres, ln, ln, ln, ln, alpha=0.5, qkv=(pq, bq)),
                        max(args.steps, 50))
    ffn_flops = 2 * (2.0 * m * 256 * hid * 2) + 2.0 * m * 256 * nqkv
    gemm_tf = ffn_flops / gemm_s / 1e12

    # ---- roofline of the fbank kernel (HBM bound): algorithmic bytes = waves in + features out ------------
    n_frames = 1 + SAMPLES // HOP
    win = _host.device_window("hann", N_FFT, N_FFT, dev)
    bank = _host.device_htk_bank(N_FFT, 0.0, float(SR // 2), N_MELS, SR, dev)
    ws = _host.workspace(lib.ma_fbank_workspace_bytes(BATCH, n_frames), dev)
    fo = torch.empty((BATCH, N_MELS, n_frames), device=dev)
    stream = _host.current_stream_ptr()

    def fbank_main_kernel():
        rc = lib.ma_fbank_db_f32(_host.ptr(x), BATCH, SAMPLES, x.stride(0), N_FFT, HOP, _host.ptr(win), 1, 1,
                                 bank.ref(), 2.0, 10.0, 1e-10, 0.0, -1.0, _host.ptr(fo), _host.ptr(ws), ws.numel(),
                                 stream)
        assert rc == 0

    fb_s = event_time(fbank_main_kernel, max(args.steps, 50))
    fb_bytes = BATCH * SAMPLES * 4 + BATCH * N_MELS * n_frames * 4  # 61 460 480 B
    fb_gbs = fb_bytes / fb_s / 1e9

    def pmc_traffic(kernel):
        """HBM bytes per launch of `kernel` from the committed PMC summary (rocprofv3 cannot run inside the bench)."""
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
                return int(json.load(fh)[kernel]["bytes"])
        except (OSError, KeyError, ValueError):
            return None

    if rank == 0:
        flops_utt = 23.12e9
        res = {
            "metric": "utterances/s (16 kHz×10 s) fbanks+Conformer fwd, 1/2/4/8 MI355X",
            "value": round(world * BATCH * args.steps / dt, 1),
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": "features.fbank (n_fft=512 hop=160 n_mels=80, batch-global top_db) on 64 x (10 s "
                                   "@16 kHz) synthetic waves per GPU -> Conformer-small encoder forward (12 blocks, "
                                   "d=256, 4 heads, ff=2048, conv k=15, eval mode) on the (64, 1000, 80) features; "
                                   "random-init weights",
                       "global_batch": BATCH * world, "frames": FRAMES,
                       "sharding": "independent utterance shards per rank, no collective",
                       "encoder_tflops": round(world * BATCH * args.steps * flops_utt / dt / 1e12, 1)},
            "roofline": {"bound": "mfma", "kernel": "ffn_packed_kernel, pair + qkv form (2 x [w_1 -> Swish -> w_2 + residual] + 4 LayerNorms + linear_q/k/v, "
                                                    "M=%d d=256 hidden=%d)" % (m, hid),
                         "achieved": round(gemm_tf, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(gemm_tf / MFMA_BF16_PEAK_TF, 4), "traffic": pmc_traffic("ffn_packed_kernel"),
                         "algorithmic_flops_per_launch": int(ffn_flops), "kernel_ms": round(gemm_s * 1e3, 5)},
            "roofline_fbank": {"bound": "hbm", "kernel": "feat512_kernel<mel>", "achieved": round(fb_gbs, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fb_gbs / HBM_PEAK_GBS, 4),
                               "traffic": pmc_traffic("feat512_kernel"), "algorithmic_bytes_per_launch": fb_bytes,
                               "kernel_ms": round(fb_s * 1e3, 5)},
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
