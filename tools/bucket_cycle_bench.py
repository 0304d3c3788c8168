#!/usr/bin/env python3
"""The training step over the shipped yaml's 16 frame buckets in the order real data brings them (a different bucket every step):
ms per step when the buckets cycle against the mean of the same shapes stepped one at a time (each at its steady state).  The
difference is what changing the batch shape costs: plan rebuilds, launch tables evicted and re-recorded, allocator churn.
    python tools/bucket_cycle_bench.py [--ctc-weight 0.3] [--rounds 6]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hybrid_soak import make_batch


def run(ctc_weight=0.3, rounds=6, settle=3, single_warm=4, single_timed=5):
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(777)
    hybrid = ctc_weight != 1.0
    model = create_asr_model(80, 4233, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12), ctc_weight=ctc_weight,
                             decoder_conf=dict(attention_heads=4, linear_units=2048, num_blocks=6, dropout_rate=0.1,
                                               positional_dropout_rate=0.1) if hybrid else None, lsm_weight=0.1 if hybrid else 0.0).to(dev)
    eng = ConformerCTCTrainStep(model, dropout_rate=0.1, positional_dropout_rate=0.1)
    fr = [144, 204, 288, 400, 512, 600, 712, 800, 912, 1024, 1112, 1200, 1400, 1600, 2000, 3000]
    bs = [40, 80, 80, 72, 72, 56, 56, 56, 40, 40, 40, 40, 24, 8, 8, 8]
    rng = np.random.RandomState(5)
    batches = [make_batch(rng, b, t, 4233, 30, dev) for b, t in zip(bs, fr)]
    order = [(7 * i) % 16 for i in range(16)]  # every bucket once per round, never two neighbours in a row
    torch.cuda.reset_peak_memory_stats()
    single = []
    for k in range(16):  # one shape at a time: steady state of each
        for _ in range(single_warm):
            eng.step(*batches[k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(single_timed):
            eng.step(*batches[k])
        torch.cuda.synchronize()
        single.append((time.perf_counter() - t0) / single_timed * 1e3)
    for _ in range(settle):  # cycling: rounds to settle (a table is recorded on a shape's second sighting)
        for k in order:
            eng.step(*batches[k])
    torch.cuda.synchronize()
    import gc

    gc_mode = os.environ.get("MA_BUCKET_GC", "")
    if gc_mode == "off":
        gc.disable()
    elif gc_mode == "freeze":
        gc.collect()
        gc.freeze()
    g0 = [st["collections"] for st in gc.get_stats()]
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    for _ in range(rounds):
        for k in order:
            eng.step(*batches[k])
    torch.cuda.synchronize()
    cyc = (time.perf_counter() - t0) / (rounds * 16) * 1e3
    g1 = [st["collections"] for st in gc.get_stats()]
    ms1 = torch.cuda.memory_stats()
    alloc = {k: ms1.get(k, 0) - ms0.get(k, 0) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}
    alloc["reserved_gb"] = round(ms1.get("reserved_bytes.all.current", 0) / 2 ** 30, 1)
    if gc_mode:
        gc.enable()
        print("gc mode %s: collections per generation during the timed rounds %s, tracked objects %d" % (gc_mode, [b - a for a, b in zip(g0, g1)], len(gc.get_objects())), file=sys.stderr)
    utts = sum(bs) * rounds
    return {"workload": "the training step (ctc_weight %.1f) over the 16 (batch, frames) buckets of conformer.yaml, a different bucket every "
                        "step (labels up to 30 tokens)" % ctc_weight,
            "ms_per_step_cycling": round(cyc, 3), "ms_per_step_one_shape_at_a_time": round(float(np.mean(single)), 3),
            "utterances_per_s_cycling": round(utts / (cyc * rounds * 16 / 1e3), 1),
            "overhead_pct": round((cyc / float(np.mean(single)) - 1) * 100, 1), "per_bucket_ms": [round(v, 2) for v in single],
            "peak_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "allocator_during_timed_rounds": alloc}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ctc-weight", type=float, default=0.3)
    ap.add_argument("--rounds", type=int, default=6)
    a = ap.parse_args()
    res = run(a.ctc_weight, a.rounds)
    res["ctc_weight"] = a.ctc_weight
    print(json.dumps(res))


if __name__ == "__main__":
    main()
