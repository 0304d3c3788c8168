#!/usr/bin/env python3
"""Micro-benchmark of the feature kernels (HIP events on the launch stream). Not the headline bench."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mindaudio_amd as ma
from mindaudio_amd import _host, _lib
from mindaudio_amd.conformer.dataset import compute_fbank_feats_batch

def timeit(fn, reps=50, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us

B = int(os.environ.get("B", 64)); N = 160000
x = torch.from_numpy((0.1*np.random.RandomState(1234).randn(B, N)).astype(np.float32)).cuda()
xk = x * 32768.0
lens = torch.full((B,), N, dtype=torch.int64, device="cuda")
lib = _lib.load()
win = _host.device_window("hann", 512, 512, x.device)
bank = _host.device_htk_bank(512, 0.0, 8000.0, 80, 16000, x.device)
T = 1001
ws = _host.workspace(lib.ma_fbank_workspace_bytes(B, T), x.device)
out = torch.empty((B, 80, T), device="cuda")
st = _host.current_stream_ptr()
def k_db(top):
    return lambda: lib.ma_fbank_db_f32(_host.ptr(x), B, N, N, 512, 160, _host.ptr(win), 1, 1, bank.ref(), 2.0, 10.0, 1e-10, 0.0, top, _host.ptr(out), _host.ptr(ws), ws.numel(), st)
def k_mel():
    return lib.ma_melspectrogram_f32(_host.ptr(x), B, N, N, 512, 160, _host.ptr(win), 1, 1, bank.ref(), 2.0, _host.ptr(out), st)
so = torch.empty((B, T, 257, 2), device="cuda")
def k_stft():
    return lib.ma_stft_f32(_host.ptr(x), B, N, N, 512, 160, _host.ptr(win), 1, 0, 0, _host.ptr(so), st)
res = {
  "fbank_main_only_us": timeit(k_db(-1.0)),
  "fbank_with_topdb_us": timeit(k_db(80.0)),
  "melspec_us": timeit(k_mel),
  "stft_us": timeit(k_stft),
  "kaldi_us": timeit(lambda: compute_fbank_feats_batch(xk, lens)),
  "py_fbank_us": timeit(lambda: ma.fbank(x, n_mels=80, n_fft=512, hop_length=160)),
  "copy_61MB_us": timeit(lambda: out.copy_(out)),
}
algo = B*N*4 + B*80*T*4
res["fbank_main_GBs"] = algo / res["fbank_main_only_us"] / 1e3
res["stft_GBs"] = (B*N*4 + B*T*257*8) / res["stft_us"] / 1e3
print(json.dumps(res))
