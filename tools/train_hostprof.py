#!/usr/bin/env python3
"""Development: where the HOST time of a training step goes (cProfile of 5 steady-state steps of the cfg-4 step).
python tools/train_hostprof.py   (GPU box)"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    dev = torch.device("cuda", 0)
    torch.manual_seed(777)
    model = create_asr_model(80, 4233, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12)).to(dev)
    eng = ConformerCTCTrainStep(model, dropout_rate=0.1, positional_dropout_rate=0.1)
    rng = np.random.RandomState(1)
    b, t = 40, 1024
    xs = torch.from_numpy(rng.randn(b, t, 80).astype(np.float32)).to(dev)
    t2 = ((t - 3) // 2 + 1 - 3) // 2 + 1
    masks = torch.ones(b, 1, t2, device=dev)
    ylens = rng.randint(5, 31, b).astype(np.int32)
    ys = np.full((b, 30), -1, np.int32)
    for i, n in enumerate(ylens):
        ys[i, :n] = rng.randint(1, 4232, n)
    cols = (xs, torch.from_numpy(ys).to(dev), None, None, None, None, masks, None, None, torch.from_numpy(ylens).to(dev), None)
    for _ in range(3):
        eng.step(*cols)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        eng.step(*cols)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(30)


main()
