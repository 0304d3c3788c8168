#!/usr/bin/env python3
"""How fast does the data side hand batches to the training step?  A synthetic AISHELL-shaped manifest (utterances of 2 ... 12 s,
16 kHz 16-bit wav files in a temporary directory), the shipped yaml's buckets / batch sizes, speed perturbation + SpecAugment on:
ms per collated batch from `create_dataset`'s iterator alone (wav reading on the host, everything else on the device) - to be held
against the ~9.5 ms the hybrid step takes.
    python tools/loader_bench.py [--utts 400]"""
import argparse
import json
import os
import sys
import tempfile
import time
import wave

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=400)
    ap.add_argument("--step", action="store_true", help="also: the hybrid training step fed by the loader, as conformer/train.py runs it")
    a = ap.parse_args()
    from mindaudio_amd.conformer.dataset import create_dataset

    rng = np.random.RandomState(0)
    tmp = tempfile.mkdtemp(prefix="ma_loader_")
    chars = [chr(0x4e00 + i) for i in range(200)]
    with open(os.path.join(tmp, "dict.txt"), "w") as fh:
        fh.write("".join("%s %d\n" % (s, i) for i, s in enumerate(["<blank>", "<unk>"] + chars + ["<sos/eos>"])))
    rows = ["id,duration,wav,transcript"]
    for i in range(a.utts):
        n = int(rng.randint(2 * 16000, 12 * 16000))
        p = os.path.join(tmp, "u%04d.wav" % i)
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes((rng.randn(n) * 3000).astype("<i2").tobytes())
        rows.append("%d,%.3f,%s,%s" % (i, n / 16000.0, p, "".join(rng.choice(chars, int(rng.randint(3, 28))))))
    with open(os.path.join(tmp, "train.csv"), "w") as fh:
        fh.write("\n".join(rows) + "\n")
    collate = dict(feature_extraction_conf=dict(feature_type="fbank", mel_bins=80, frame_shift=10, frame_length=25, using_pitch=False),
                   feature_dither=0.0, use_speed_perturb=True, use_spec_aug=True,
                   spec_aug_conf=dict(warp_for_time=False, num_t_mask=2, num_f_mask=2, prop_mask_t=0.1, prop_mask_f=0.1, max_t=50, max_f=10,
                                      max_w=80))
    dsc = dict(max_length=3000, min_length=0, token_max_length=30, token_min_length=1, batch_type="bucket",
               frame_bucket_limit="144, 204, 288, 400, 512, 600, 712, 800, 912, 1024, 1112, 1200, 1400, 1600, 2000, 3000",
               batch_bucket_limit="40, 80, 80, 72, 72, 56, 56, 56, 40, 40, 40, 40, 24, 8, 8, 8", batch_factor=1, shuffle=True)
    _, ds = create_dataset(os.path.join(tmp, "train.csv"), os.path.join(tmp, "dict.txt"), collate, dsc, number_workers=1)
    n_b = 0
    for cols in ds:  # first epoch: file cache, kernels warm
        n_b += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    utts = 0
    for cols in ds:
        utts += cols[0].shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"batches_per_epoch": n_b, "ms_per_batch_loader_alone": round(dt / n_b * 1e3, 2), "utterances_per_s_loader_alone": round(utts / dt, 1),
           "mean_batch": round(utts / n_b, 1)}
    if a.step:
        # the training script's loop (conformer/train.py): loader -> hybrid step -> the step's host read, batch after batch
        from mindaudio_amd.conformer.asr_model import create_asr_model
        from mindaudio_amd.train.engine import ConformerCTCTrainStep

        dev = torch.device("cuda", 0)
        torch.manual_seed(777)
        model = create_asr_model(80, 204, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12), ctc_weight=0.3,
                                 decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                                                   positional_dropout_rate=0.1), lsm_weight=0.1).to(dev)
        eng = ConformerCTCTrainStep(model, dropout_rate=0.1, positional_dropout_rate=0.1)
        for _ in range(3):  # shapes seen, tables recorded
            for cols in ds:
                float(eng.step(*cols)[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(3):
            for cols in ds:
                float(eng.step(*cols)[0])
                n += 1
        torch.cuda.synchronize()
        res["ms_per_step_loader_then_step"] = round((time.perf_counter() - t0) / n * 1e3, 2)
        # ... and as conformer/train.py runs it since round 6: the next batch collated between enqueue_step and finish_step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for _ in range(3):
            it = iter(ds)
            cols = next(it, None)
            while cols is not None:
                pending = eng.enqueue_step(*cols)
                cols = next(it, None)
                float(eng.finish_step(*pending)[0])
                n += 1
        torch.cuda.synchronize()
        res["ms_per_step_pipelined"] = round((time.perf_counter() - t0) / n * 1e3, 2)
        kept = [cols for cols in ds]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            for cols in kept:
                float(eng.step(*cols)[0])
        torch.cuda.synchronize()
        res["ms_per_step_on_resident_batches"] = round((time.perf_counter() - t0) / (3 * len(kept)) * 1e3, 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
