#!/usr/bin/env python3
"""cfg 2 CPU baselines: the oracle's fbank on the host cores, in the two flavours the survey asks for —
(R) reference-faithful cost structure (float64 framing loop + per-column rFFT, oracle.fbank_ref_cost) and (V) vectorised NumPy
(oracle.fbank) — each as one process and as multiprocessing.Pool(min(8, nproc)) over utterances (the reference's own
parallelism, examples/conformer/dataset.py:449,479).  Baseline only: nothing here is a product path."""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import speech_features as O

KW = dict(n_mels=80, n_fft=512, hop_length=160)


def _r(x):
    return O.fbank_ref_cost(x[None], **KW).shape


def _v(x):
    return O.fbank(x[None], **KW).shape


def main():
    n = int(os.environ.get("N", 16))
    x = (0.1 * np.random.RandomState(1234).randn(n, 160000)).astype(np.float32)
    rows = [x[i] for i in range(n)]
    nproc = os.cpu_count()
    out = {"cores": nproc, "utterances": n}
    for name, fn in (("R", _r), ("V", _v)):
        t0 = time.perf_counter()
        for r in rows:
            fn(r)
        out[name + "_1proc_utt_per_s"] = round(n / (time.perf_counter() - t0), 2)
        with mp.Pool(min(8, nproc)) as pool:
            pool.map(fn, rows[:2])
            t0 = time.perf_counter()
            pool.map(fn, rows)
            out[name + "_pool8_utt_per_s"] = round(n / (time.perf_counter() - t0), 2)
    print(out)


if __name__ == "__main__":
    main()
