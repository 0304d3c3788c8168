#!/usr/bin/env python3
"""bench.py — headline benchmark of the mindaudio hot path on MI355X.

One "step" = one pass of the hot path over one synthetic batch (BASELINE.json configs[1]):
features.fbank on 64 x (10 s @ 16 kHz) float32 waves, n_fft=512 hop=160 n_mels=80, inputs resident in
HBM before the timed region.  N GPUs = N independent shards of 64 utterances each (weak scaling, no
data-path collective: the reference's batch-global top_db floor is per call, i.e. per rank —
SURVEY §8e).  Prints ONE JSON line on rank 0.
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_batch(seed):
    return (0.1 * np.random.RandomState(seed).randn(BATCH, SAMPLES)).astype(np.float32)


def cpu_baseline(budget_s=15.0):
    """Oracle flavour (R) — the reference's own cost structure (float64 framing loop + per-column rFFT +
    dense mel + amplitude_to_dB) — single process, on a bounded sample of the same workload."""
    from oracle import speech_features as O

    x = synth_batch(1234)
    n_utt = 4
    t0 = time.perf_counter()
    O.fbank_ref_cost(x[:n_utt], n_mels=N_MELS, n_fft=N_FFT, sample_rate=SR, hop_length=HOP)
    dt = time.perf_counter() - t0
    reps, total, done = 1, dt, n_utt
    while total < budget_s and reps < 8:
        t0 = time.perf_counter()
        O.fbank_ref_cost(x[:n_utt], n_mels=N_MELS, n_fft=N_FFT, sample_rate=SR, hop_length=HOP)
        total += time.perf_counter() - t0
        done += n_utt
        reps += 1
    return {"value": round(done / total, 3), "unit": "utterances/s", "cores": 1, "kind": "port",
            "sample": "%d x fbank_ref_cost on %d utterances (10 s @16 kHz) of the cfg-2 batch, NumPy float64, "
                      "1 process" % (reps, n_utt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
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
    from mindaudio_amd import _host, _lib

    lib = _lib.load()
    x = torch.from_numpy(synth_batch(1234 + rank)).to(dev)
    kw = dict(n_mels=N_MELS, n_fft=N_FFT, hop_length=HOP)

    def step():
        return ma.fbank(x, **kw)

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
    n_frames = out.shape[-1]

    # ---- roofline of the dominant kernel (feat512_kernel<mel>), HIP events on the launch stream ----------
    win = _host.device_window("hann", N_FFT, N_FFT, dev)
    bank = _host.device_htk_bank(N_FFT, 0.0, float(SR // 2), N_MELS, SR, dev)
    ws = _host.workspace(lib.ma_fbank_workspace_bytes(BATCH, n_frames), dev)
    o2 = torch.empty_like(out)
    stream = _host.current_stream_ptr()

    def main_kernel_only():
        rc = lib.ma_fbank_db_f32(_host.ptr(x), BATCH, SAMPLES, x.stride(0), N_FFT, HOP, _host.ptr(win), 1, 1,
                                 bank.ref(), 2.0, 10.0, 1e-10, 0.0, -1.0, _host.ptr(o2), _host.ptr(ws),
                                 ws.numel(), stream)
        assert rc == 0

    for _ in range(10):
        main_kernel_only()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(args.steps, 50)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        main_kernel_only()
    ev1.record()
    torch.cuda.synchronize()
    kern_ms = ev0.elapsed_time(ev1) / reps
    algo_bytes = BATCH * SAMPLES * 4 + BATCH * N_MELS * n_frames * 4  # 61 460 480 B
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9

    if rank == 0:
        res = {
            "metric": "utterances/s (16 kHz x 10 s) fbanks, 1/2/4/8 MI355X",
            "value": round(world * BATCH * args.steps / dt, 1),
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "batched fbanks: 64 x (10 s @16 kHz) per GPU, n_fft=512 hop=160 n_mels=80 "
                                   "(BASELINE configs[1]); features.fbank = melspectrogram + amplitude_to_dB "
                                   "with batch-global top_db",
                       "global_batch": BATCH * world, "samples_per_utt": SAMPLES, "n_frames": int(n_frames),
                       "sharding": "independent utterance shards per rank, no collective"},
            "roofline": {"bound": "hbm", "kernel": "feat512_kernel<mel>", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": None, "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel_ms": round(kern_ms, 5)},
        }
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
