#!/usr/bin/env python3
"""The evaluation forward (Conformer-small encoder + CTC greedy search) over the yaml's 16 buckets, a different bucket every call,
against each shape called alone: what changing the batch shape costs the inference path.
    python tools/eval_cycle_bench.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from mindaudio_amd.conformer.asr_model import CTCGreedySearch, create_asr_model, ctc_greedy_search

    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    model = create_asr_model(80, 4233, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12)).to(dev).eval()
    net = CTCGreedySearch(model)
    fr = [144, 204, 288, 400, 512, 600, 712, 800, 912, 1024, 1112, 1200, 1400, 1600, 2000, 3000]
    bs = [40, 80, 80, 72, 72, 56, 56, 56, 40, 40, 40, 40, 24, 8, 8, 8]
    rng = np.random.RandomState(2)
    batches = []
    for b, t in zip(bs, fr):
        xs = torch.from_numpy(rng.randn(b, t, 80).astype(np.float32)).to(dev)
        lens = rng.randint(int(0.6 * t), t + 1, b)
        lens[0] = t
        m = torch.zeros(b, 1, t, device=dev)
        for i, n in enumerate(lens):
            m[i, 0, :n] = 1
        batches.append((xs, m))
    call = lambda k: net._search(*batches[k])  # noqa: E731  (encoder + CTC head + device greedy search, no host read)
    single = []
    for k in range(16):
        for _ in range(3):
            call(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            call(k)
        torch.cuda.synchronize()
        single.append((time.perf_counter() - t0) / 10 * 1e3)
    order = [(7 * i) % 16 for i in range(16)]
    for k in order:
        call(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        for k in order:
            call(k)
    torch.cuda.synchronize()
    cyc = (time.perf_counter() - t0) / (8 * 16) * 1e3
    print(json.dumps({"ms_per_call_cycling": round(cyc, 3), "ms_per_call_one_shape_at_a_time": round(float(np.mean(single)), 3),
                      "overhead_pct": round((cyc / float(np.mean(single)) - 1) * 100, 1), "per_bucket_ms": [round(v, 2) for v in single],
                      "utterances_per_s_cycling": round(sum(bs) * 8 / (cyc * 8 * 16 / 1e3), 1)}))


if __name__ == "__main__":
    main()
