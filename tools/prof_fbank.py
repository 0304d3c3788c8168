#!/usr/bin/env python3
"""Run the cfg-2 fbank main kernel a few times (target of rocprofv3 passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mindaudio_amd import _host, _lib
B, N, T = 64, 160000, 1001
x = torch.from_numpy((0.1*np.random.RandomState(1234).randn(B, N)).astype(np.float32)).cuda()
lib = _lib.load()
win = _host.device_window("hann", 512, 512, x.device)
bank = _host.device_htk_bank(512, 0.0, 8000.0, 80, 16000, x.device)
ws = _host.workspace(lib.ma_fbank_workspace_bytes(B, T), x.device)
out = torch.empty((B, 80, T), device="cuda")
st = _host.current_stream_ptr()
mode = sys.argv[1] if len(sys.argv) > 1 else "fbank"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
so = torch.empty((B, T, 257, 2), device="cuda") if mode == "stft" else None
for _ in range(n):
    if mode == "fbank":
        rc = lib.ma_fbank_db_f32(_host.ptr(x), B, N, N, 512, 160, _host.ptr(win), 1, 1, bank.ref(), 2.0, 10.0, 1e-10, 0.0, 80.0, _host.ptr(out), _host.ptr(ws), ws.numel(), st)
    else:
        rc = lib.ma_stft_f32(_host.ptr(x), B, N, N, 512, 160, _host.ptr(win), 1, 0, 0, _host.ptr(so), st)
    assert rc == 0
torch.cuda.synchronize()
