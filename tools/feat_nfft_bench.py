#!/usr/bin/env python3
"""features.fbank on the cfg-2 input (64 x 10 s) for n_fft = 512 (radix 16 x 16), 400 (radix 25 x 8: the reference's default,
features.py:201) and a size on the exact-f32 MFMA DFT coverage path; hop 160, 80 mel.  HIP events on the launch stream."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mindaudio_amd as ma
B = int(os.environ.get("B", 64))
x = torch.from_numpy((0.1 * np.random.RandomState(1234).randn(B, 160000)).astype(np.float32)).cuda()
def timeit(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
res = {}
for n_fft in (512, 400, 256):
    res["fbank_n_fft_%d_us" % n_fft] = round(timeit(lambda: ma.fbank(x, n_mels=80, n_fft=n_fft, hop_length=160)), 1)
    res["stft_n_fft_%d_us" % n_fft] = round(timeit(lambda: ma.stft(x, n_fft=n_fft, hop_length=160)), 1)
res["fbank_default_args_us"] = round(timeit(lambda: ma.fbank(x)), 1)  # n_fft 400, hop 200, 40 mel
print(json.dumps(res))
