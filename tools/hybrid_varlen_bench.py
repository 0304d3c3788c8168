#!/usr/bin/env python3
"""The hybrid training step when the longest transcript changes from batch to batch (what real data does): one encoder shape
(40 x 1024 frames), label widths cycling through 21 values - the decoder's launch table replays only for the recorded width, every
other step walks the decoder from Python beside the replayed encoder.  Prints ms per step against the fixed-width figure.
    python tools/hybrid_varlen_bench.py [--steps 63]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hybrid_soak import make_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=63)
    ap.add_argument("--plans-kept", type=int, default=0, help="override the engine's cache of LayerNorm partial-sum plans (1: round 6 before the fix)")
    a = ap.parse_args()
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    dev = torch.device("cuda", 0)
    torch.manual_seed(777)
    model = create_asr_model(80, 4233, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12), ctc_weight=0.3,
                             decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                                               positional_dropout_rate=0.1), lsm_weight=0.1).to(dev)
    eng = ConformerCTCTrainStep(model, dropout_rate=0.1, positional_dropout_rate=0.1)
    if a.plans_kept:
        eng._DEC_LN_PLANS_KEPT = a.plans_kept
    rng = np.random.RandomState(3)
    out = {}
    for name, widths in (("fixed width 30", [30]), ("21 widths 20..40", list(range(20, 41)))):
        batches = [make_batch(rng, 40, 1024, 4233, w, dev) for w in widths]
        for k in range(2 * len(batches) + 6):  # every width seen once or twice, the tables recorded
            eng.step(*batches[k % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            eng.step(*batches[(k * 5) % len(batches)])
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / a.steps * 1e3, 3)
    print(json.dumps({"ms_per_step": out}))


if __name__ == "__main__":
    main()
