"""Phase timeline of subsample_fused_kernel from a -DMA_SF_PROF build of subsample_fused.hip
(SRC=subsample_fused.hip bash tools/ffn_variants.sh build "prof:-DMA_SF_PROF"; run with MINDAUDIO_AMD_LIB=.../variants/prof.so):
s_memtime stamps of consumer wave 0 and producer wave 4 of three workgroups, in shader-clock cycles since the workgroup's entry."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mindaudio_amd import _lib, ops

lib = _lib.load()
B, T, F = 64, 1000, 80
feats = torch.randn(B, F, T + 1, device="cuda").transpose(1, 2)[:, :T]
w1 = (torch.randn(256, 9, device="cuda") * 0.3).contiguous()
b1 = torch.randn(256, device="cuda") * 0.1
w2 = (torch.randn(256, 3, 3, 256, device="cuda") / 48).bfloat16()
b2 = torch.randn(256, device="cuda")
spk = ops.subsample_fused_pack(w1, w2, F)
fn = lambda: ops.subsample_fused(feats, spk, b1, b2)
for _ in range(5):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
print("launch %.1f us (back to back, instrumented build)" % (e0.elapsed_time(e1) / 20 * 1e3))
lib.ma_debug_sf_prof.argtypes = [ctypes.c_void_p]
acc = []
for it in range(10):
    fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 192)()
    assert lib.ma_debug_sf_prof(buf) == 0
    acc.append(np.array(buf[:], dtype=np.int64).reshape(2, 3, 32))
a = np.median(np.stack(acc), axis=0)
for role, name in ((0, "consumer wave 0"), (1, "producer wave 4")):
    print("== %s: cycles since the wave's entry, workgroups 0 / 1200 / 2300" % name)
    labels = {0: "entry", 1: "rows staged", 2: "X built", 3: "patch 0 built", 20: "loop end / last barrier", 21: "end"}
    for k in range(8):
        labels[4 + 2 * k] = ("chunk %d start" % k) if role == 0 else ("patch %d start" % (k + 1))
        labels[5 + 2 * k] = ("chunk %d end" % k) if role == 0 else ("patch %d built" % (k + 1))
    prev = None
    for k in sorted(labels):
        v = a[role, :, k] - a[role, :, 0]
        if a[role, 0, k] == 0:
            continue
        d = "" if prev is None else "  (+%s)" % " / ".join("%6d" % int(x) for x in (v - prev))
        print("%-24s %s%s" % (labels[k], " / ".join("%7d" % int(x) for x in v), d))
        prev = v
