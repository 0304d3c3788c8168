#!/usr/bin/env python3
"""Phase timeline of relpos_attention_kernel from a -DMA_ATT_PROF build (tools/lib_variant.sh attprof "-DMA_ATT_PROF"
conformer_kernels.hip; MINDAUDIO_AMD_LIB=mindaudio_amd/lib/variants/attprof.so): wall_clock64 stamps (100 MHz) of wave 0 of three
workgroups (first, middle, last).  B = 64, T' = 249, 4 heads."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mindaudio_amd import _lib, ops
lib = _lib.load()
lib.ma_debug_att_prof.argtypes = [ctypes.c_void_p]
B, T = int(os.environ.get("B", 64)), 249
qkv = torch.randn(B * T, 768, device="cuda").bfloat16(); pos = torch.randn(T, 256, device="cuda").bfloat16()
u, v = torch.randn(4, 64, device="cuda") * 0.3, torch.randn(4, 64, device="cuda") * 0.3
mask = torch.ones(B, T, device="cuda")
out = torch.empty(B * T, 256, device="cuda", dtype=torch.bfloat16)
fn = lambda: ops.relpos_attention(qkv, pos, u, v, mask, B, T, 4, 64, out=out)
for _ in range(10): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
print("launch %.1f us (back to back, instrumented build)" % (e0.elapsed_time(e1) / 50 * 1e3))
names = ["entry", "Q' built, tile 0 issued", "tile 0 published", "tile 0 consumed", "tile 1 published", "tile 1 consumed", "tile 2 published",
         "tile 2 consumed", "tile 3 published", "tile 3 consumed", "stores issued", "stores retired"]
acc = {}
N = 20
for it in range(N):
    fn(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 48)()
    assert lib.ma_debug_att_prof(buf) == 0
    a = np.array(buf[:], dtype=np.int64).reshape(3, 16)
    t0 = a[:, 0].min()
    for w in range(3):
        for k in range(12):
            acc.setdefault((w, k), []).append((a[w, k] - t0) / 100.0)
print("%-26s %14s %14s %14s   (us since the first start; median of %d; +delta)" % ("phase", "first wg", "middle wg", "last wg", N))
prev = [0, 0, 0]
for k in range(12):
    med = [float(np.median(acc[(w, k)])) for w in range(3)]
    print("%-26s " % names[k] + " ".join("%6.2f(+%5.2f)" % (med[w], med[w] - prev[w]) for w in range(3)))
    prev = med
