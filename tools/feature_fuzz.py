#!/usr/bin/env python3
"""Random shapes and arguments through the feature entry points against the oracle (test infrastructure: this tool is a checker, like
tests/): stft, fbank (melspectrogram + dB), the Kaldi fbank of the Conformer loader on ragged batches, istft round trips, mfcc,
compute_deltas, resample.  The parity tests pin chosen cases; this walks the space between them.
    python tools/feature_fuzz.py [--cases 300] [--seed 0]          (prints one JSON line; exit code 1 on a failure)"""
import argparse
import json
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def frame_rel(got, want):
    scale = np.maximum(np.abs(want).max(axis=-2, keepdims=True), 1e-30)
    return float((np.abs(got - want) / scale).max())


def run(cases=300, seed=0):
    import scipy.signal

    import mindaudio_amd as ma
    from mindaudio_amd.conformer.dataset import compute_fbank_feats_batch
    from mindaudio_amd.data.processing import resample_batch
    from oracle import speech_features as O

    rng = np.random.RandomState(seed)
    counts, worst, failures = {}, {}, []

    def note(kind, err, tol, desc):
        counts[kind] = counts.get(kind, 0) + 1
        worst[kind] = max(worst.get(kind, 0.0), err / tol)
        if not err <= tol:
            failures.append((kind, desc, err, tol))

    for case in range(cases):
        kind = ("stft", "fbank", "kaldi", "istft", "mfcc", "deltas", "resample")[case % 7]
        try:
            if kind == "stft":
                n_fft = int(rng.choice([128, 256, 400, 512, 1024]))
                win = int(rng.choice([n_fft, n_fft, max(16, n_fft // 2), max(16, (n_fft * 25) // 32)]))
                hop = int(rng.randint(1, n_fft))
                center = bool(rng.randint(2))
                n = int(rng.randint(n_fft, 6 * n_fft + 40))  # (shorter signals raise, as the reference does: tests/test_features_gpu.py)
                kw = dict(n_fft=n_fft, win_length=win, hop_length=hop, center=center, window=str(rng.choice(["hann", "hamming"])),
                          pad_mode=str(rng.choice(["constant", "reflect", "edge"])))
                shape = (n,) if rng.randint(3) == 0 else (int(rng.randint(1, 5)), n)
                x = (rng.randn(*shape) * 10 ** rng.uniform(-3, 1)).astype(np.float32)
                if kw["pad_mode"] == "reflect" and center and n <= n_fft // 2:
                    continue
                got, want = ma.stft(x, **kw), O.stft_vec(x, **kw)
                assert got.shape == want.shape, (got.shape, want.shape)
                note(kind, frame_rel(got, want), 2e-6 if n_fft in (512, 400) else 1e-5, kw | {"shape": shape})
            elif kind == "fbank":
                n_fft = int(rng.choice([400, 512]))
                hop = int(rng.choice([n_fft // 2, 160, 128, 200, int(rng.randint(40, n_fft))]))
                n_mels = int(rng.choice([23, 40, 64, 80]))
                n = int(rng.randint(n_fft, 20000))
                shape = [(n,), (int(rng.randint(1, 6)), n), (2, int(rng.randint(1, 3)), n)][rng.randint(3)]
                x = (rng.randn(*shape) * 10 ** rng.uniform(-3, 0)).astype(np.float32)
                kw = dict(n_mels=n_mels, n_fft=n_fft, hop_length=hop)
                got, want = ma.fbank(x, **kw), O.fbank(x, **kw)
                assert got.shape == want.shape, (got.shape, want.shape)
                live = want > want.max() - 70.0  # (within 10 dB of the floor the float32 noise floor of the frame decides)
                note(kind, float(np.abs(got - want)[live].max()), 5e-3, kw | {"shape": shape})
            elif kind == "kaldi":
                b = int(rng.randint(1, 7))
                lens = [int(rng.randint(400, 40000)) for _ in range(b)]  # (below one frame: tests/test_features_gpu.py, from goldens)
                mel_bin = int(rng.choice([40, 80]))
                host = np.zeros((b, max(lens)), np.float32)
                for i, n in enumerate(lens):
                    host[i, :n] = rng.randn(n) * 10 ** rng.uniform(1, 4)
                feats, nfr = compute_fbank_feats_batch(host, lens, sample_rate=16000, frame_len=25, frame_shift=10, mel_bin=mel_bin)
                feats, nfr = feats.cpu().numpy(), nfr.cpu().numpy()
                err = 0.0
                for i, n in enumerate(lens):
                    want = O.compute_fbank_feats(host[i, :n].astype(np.float64), 16000, 25, 10, mel_bin)
                    assert int(nfr[i]) == want.shape[0], (int(nfr[i]), want.shape)
                    if want.shape[0]:
                        # as tests/test_features_gpu.py::_check_ln: bins within 1e-4 of the frame's largest energy at 2e-3, the rest
                        # (the frame's float32 noise floor decides) at 5e-2
                        d = np.abs(feats[i, :want.shape[0]] - want)
                        e = np.exp(want)
                        strong = e >= 1e-4 * e.max(axis=-1, keepdims=True)
                        err = max(err, float(d[strong].max()), float(d.max()) * (2e-3 / 5e-2))
                        assert not feats[i, want.shape[0]:].any()  # zero rows behind the utterance
                note(kind, err, 2e-3, dict(lens=lens, mel_bin=mel_bin))
            elif kind == "istft":
                n_fft = int(rng.choice([256, 400, 512]))
                hop = int(rng.choice([n_fft // 4, n_fft // 2, 160 if n_fft >= 400 else 64]))
                n = int(rng.randint(2 * n_fft, 12000))
                x = rng.randn(n).astype(np.float32)
                S = ma.stft(x, n_fft=n_fft, hop_length=hop)
                y = ma.istft(S, n_fft=n_fft, hop_length=hop, length=n)
                want = O.istft(O.stft_vec(x, n_fft=n_fft, hop_length=hop), n_fft=n_fft, hop_length=hop, length=n)
                # the last hop samples divide by the tail of the window-sum-square envelope, which goes to zero at the last frame's
                # edge: there the float64 oracle's own round trip is off by 4e-5 (n = 58 * 128 - 1) - a looser bound for that tail
                d = np.abs(np.asarray(y) - want)
                scale = max(1.0, float(np.abs(want).max()))
                note(kind, max(float(d[:-hop].max()), float(d[-hop:].max()) * (2e-5 / 5e-3)), 2e-5 * scale, dict(n_fft=n_fft, hop=hop, n=n))
            elif kind == "mfcc":
                n = int(rng.randint(4000, 20000))
                b = int(rng.randint(1, 4))
                x = (rng.randn(b, n) * 0.1).astype(np.float32)
                kw = dict(n_mels=int(rng.choice([23, 40])), n_mfcc=int(rng.choice([13, 20])), deltas=bool(rng.randint(2)), context=bool(rng.randint(2)))
                got, want = np.asarray(ma.mfcc(x, **kw)), O.mfcc(x, **kw)
                assert got.shape == want.shape, (got.shape, want.shape)
                note(kind, float(np.abs(got - want).max()), 2e-2, kw | {"shape": x.shape})
            elif kind == "deltas":
                shape = (int(rng.randint(1, 4)), int(rng.randint(3, 90)), int(rng.randint(1, 400)))
                x = rng.randn(*shape).astype(np.float32)
                win = int(rng.choice([3, 5, 7, 9]))
                got, want = np.asarray(ma.compute_deltas(x, win_length=win)), O.compute_deltas(x, win_length=win)
                note(kind, float(np.abs(got - want).max()), 1e-5 * max(1.0, float(np.abs(want).max())), dict(shape=shape, win=win))
            else:
                import torch

                b = int(rng.randint(1, 5))
                n_in = [int(rng.randint(2000, 90000)) for _ in range(b)]
                speed = [float(rng.choice([0.9, 1.1, 0.95, 1.234])) for _ in range(b)]
                n_out = [int(np.ceil(n / s)) for n, s in zip(n_in, speed)]
                host = np.zeros((b, max(n_in)), np.float32)
                for i, n in enumerate(n_in):
                    host[i, :n] = rng.randn(n)
                y = resample_batch(torch.from_numpy(host).cuda(), n_in, n_out).cpu().numpy()
                err = 0.0
                for i in range(b):
                    want = scipy.signal.resample(host[i, :n_in[i]].astype(np.float64), n_out[i])
                    err = max(err, float(np.abs(y[i, :n_out[i]] - want).max() / np.abs(want).max()))
                    if y.shape[1] > n_out[i]:
                        err = max(err, float(np.abs(y[i, n_out[i]:]).max()))  # padded to the batch's longest row with zeros
                note(kind, err, 2e-5, dict(n_in=n_in, n_out=n_out))
        except Exception as e:  # noqa: BLE001  (a crash is a finding too)
            failures.append((kind, "case %d raised %s: %s" % (case, type(e).__name__, str(e)[:300]), float("nan"), 0.0))
            traceback.print_exc()
    return {"cases": counts, "worst_error_over_tolerance": {k: round(v, 3) for k, v in worst.items()},
            "failures": [(k, str(d)[:400], e, t) for k, d, e, t in failures[:12]], "n_failures": len(failures)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    res = run(a.cases, a.seed)
    print(json.dumps(res))
    sys.exit(1 if res["n_failures"] else 0)


if __name__ == "__main__":
    main()
