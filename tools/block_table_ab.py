#!/usr/bin/env python3
"""cfg-4 training step (bench.py's train_leg: 40 x 1024 x 80, 12 blocks) with the block launch table on and off: step time, host time
to enqueue one step behind a busy GPU, and a hash of the trained masters (must agree: the table re-issues the walked step's calls).
    python tools/block_table_ab.py [--hybrid] [--steps 10]"""
import argparse
import gc
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--hybrid", action="store_true")
    args = ap.parse_args()
    import torch

    import bench
    from mindaudio_amd.train import engine

    dev = torch.device("cuda:0")
    init = engine.ConformerCTCTrainStep.__init__
    out = {}
    for tables in (True, False, True, False):
        def patched(self, *a, _t=tables, **k):
            init(self, *a, **k)
            self.block_tables = self.block_tables and _t
        engine.ConformerCTCTrainStep.__init__ = patched
        try:
            res = bench.train_leg(0, 1, dev, None, args.steps, 3, torch.cuda.synchronize, digest=True,
                                  ctc_weight=0.3 if args.hybrid else 1.0)
        finally:
            engine.ConformerCTCTrainStep.__init__ = init
        gc.collect()  # (an engine is a reference cycle: without this the previous arm's tables stay allocated under the next arm)
        torch.cuda.empty_cache()
        out.setdefault("tables_on" if tables else "tables_off", []).append(
            {k: res[k] for k in ("ms_per_step", "host_enqueue_ms", "masters_sha16", "last_loss")})
    out["masters_equal"] = len({r["masters_sha16"] for v in out.values() for r in v}) == 1
    print(json.dumps(out))


if __name__ == "__main__":
    main()
