#!/usr/bin/env python3
"""Soak of the hybrid (0.3 CTC + 0.7 attention) training step over changing bucket shapes and label lengths - what a real epoch does to
the launch tables (record / replay / eviction per batch shape, another label length than the recorded one, table byte budget):
every loss finite, the device memory bounded, and at the end the engine's weights equal those of an engine that walked every step
from Python (block_tables off) on the same batches - bit for bit.

    python tools/hybrid_soak.py [--steps 120] [--blocks 12]"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def make_batch(rng, b, t, vocab, lmax, dev):
    xs = torch.from_numpy(rng.randn(b, t, 80).astype(np.float32))
    lens = rng.randint(int(0.6 * t), t + 1, b)
    lens[0] = t
    t2 = ((t - 3) // 2 + 1 - 3) // 2 + 1
    masks = torch.zeros(b, 1, t2)
    for i, n in enumerate(lens):
        masks[i, 0, :(n - 1) // 4] = 1
        xs[i, n:] = 0
    ylens = rng.randint(3, lmax + 1, b).astype(np.int32)
    ylens[0] = lmax
    ys = np.full((b, lmax), -1, np.int32)
    for i, n in enumerate(ylens):
        ys[i, :n] = rng.randint(1, vocab - 1, n)
    eos = vocab - 1
    ys_in = np.full((b, lmax + 1), eos, np.int32)
    ys_out = np.full((b, lmax + 1), -1, np.int32)
    ys_m = np.zeros((b, 1, lmax + 1), np.float32)
    for i, n in enumerate(ylens):
        ys_in[i, 1:n + 1] = ys[i, :n]
        ys_out[i, :n] = ys[i, :n]
        ys_out[i, n] = eos
        ys_m[i, 0, :n + 1] = 1
    ys_sub = (ys_m.astype(bool) & np.tril(np.ones((lmax + 1, lmax + 1), bool))[None]).astype(np.float32)
    t_ = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    return (xs.to(dev), t_(ys), t_(ys_in), t_(ys_out), None, None, masks.to(dev), t_(ys_sub), t_(ys_m), t_(ylens), None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--blocks", type=int, default=12)
    ap.add_argument("--vocab", type=int, default=4233)
    ap.add_argument("--ctc-weight", type=float, default=0.3, help="1.0: the pure-CTC step (no decoder)")
    ap.add_argument("--len-norm", action="store_true", help="length_normalized_loss (asr_model.py:61): the decoder is always walked")
    ap.add_argument("--yaml-buckets", action="store_true", help="the 16 (batch, frames) buckets of conformer.yaml, labels up to 30 tokens")
    ap.add_argument("--only", default="", help="tables | walked: one engine only (no comparison)")
    ap.add_argument("--verbose", action="store_true", help="synchronise and print after every step")
    a = ap.parse_args()
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    dev = torch.device("cuda", 0)

    def build(tables):
        torch.manual_seed(777)
        model = create_asr_model(80, a.vocab, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=a.blocks),
                                 ctc_weight=a.ctc_weight,
                                 decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                                                   positional_dropout_rate=0.1) if a.ctc_weight != 1.0 else None,
                                 lsm_weight=0.1 if a.ctc_weight != 1.0 else 0.0, length_normalized_loss=a.len_norm).to(dev)
        eng = ConformerCTCTrainStep(model, base_lr=5e-4, warmup_steps=50, dropout_rate=0.1, positional_dropout_rate=0.1, seed=11)
        if not tables:
            eng.block_tables = False
        return eng

    # the buckets of conformer.yaml scaled to one GPU: (batch, frames, longest label)
    shapes = [(40, 1024, 30), (24, 1536, 42), (64, 640, 18), (40, 1024, 23), (12, 3000, 60), (40, 1024, 30)]
    if a.yaml_buckets:
        fr = [144, 204, 288, 400, 512, 600, 712, 800, 912, 1024, 1112, 1200, 1400, 1600, 2000, 3000]
        bs = [40, 80, 80, 72, 72, 56, 56, 56, 40, 40, 40, 40, 24, 8, 8, 8]
        shapes = [(b, t, 12 + (7 * i) % 19) for i, (b, t) in enumerate(zip(bs, fr))]
    rng = np.random.RandomState(5)
    # each shape recurs (a table is recorded on its second sighting and replayed from the third on)
    order = [shapes[(i // 3 + i) % len(shapes)] for i in range(a.steps)]
    batches = [make_batch(rng, b, t, a.vocab, lm, dev) for (b, t, lm) in shapes]
    idx = [shapes.index(s) for s in order]
    out = {}
    for name, tables in (("tables", True), ("walked", False)):
        if a.only and a.only != name:
            continue
        eng = build(tables)
        torch.cuda.reset_peak_memory_stats()
        losses, t0 = [], time.time()
        for k in idx:
            if a.verbose:
                print(name, "step", len(losses), "shape", shapes[k], flush=True)
            loss, cond, scale, overflow, lr = eng.step(*batches[k])
            if a.verbose:
                torch.cuda.synchronize()
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert all(math.isfinite(v) for v in losses), losses
        eng.sync_to_module()
        out[name] = dict(losses=losses, seconds=round(time.time() - t0, 1), peak_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                         master=eng.fp.master.clone())
        del eng
        torch.cuda.empty_cache()
    if a.only:
        print(json.dumps({k: v for k, v in out[a.only].items() if k != "master"})[:600])
        return
    same = torch.equal(out["tables"]["master"], out["walked"]["master"])
    loss_same = out["tables"]["losses"] == out["walked"]["losses"]
    print(json.dumps(dict(steps=a.steps, shapes=shapes, first_loss=out["tables"]["losses"][0], last_loss=out["tables"]["losses"][-1],
                          weights_bit_identical=same, losses_identical=loss_same,
                          seconds_tables=out["tables"]["seconds"], seconds_walked=out["walked"]["seconds"],
                          peak_gb_tables=out["tables"]["peak_gb"], peak_gb_walked=out["walked"]["peak_gb"])))
    assert same and loss_same


if __name__ == "__main__":
    main()
