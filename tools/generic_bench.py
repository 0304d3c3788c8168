"""Times the generic-n_fft feature path (features_generic.hip) next to the 512 fast path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mindaudio_amd as ma

x = torch.from_numpy((0.1 * np.random.RandomState(0).randn(64, 160000)).astype(np.float32)).cuda()
for n_fft in (400, 512, 256, 1024):
    kw = dict(n_mels=80, n_fft=n_fft, hop_length=160)
    for _ in range(3):
        ma.fbank(x, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        ma.fbank(x, **kw)
    e1.record(); torch.cuda.synchronize()
    print("fbank n_fft=%d: %.1f us" % (n_fft, e0.elapsed_time(e1) / 20 * 1e3))
