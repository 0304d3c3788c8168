"""Phase timeline of the one-launch feed-forward training module (forward and backward) from a -DFT_PROF build
(SRC=ffn_train.hip bash tools/ffn_variants.sh build "prof:-DFT_PROF"; MINDAUDIO_AMD_LIB=mindaudio_amd/lib/variants/prof.so):
wall_clock64 stamps (100 MHz) of wave 0 of three workgroups."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mindaudio_amd import _lib, ops
from mindaudio_amd.train import kernels as K

lib = _lib.load()
m, d, hid, p, seed = 10200, 256, 2048, 0.1, 7
g = torch.Generator().manual_seed(0)
bf = lambda x: x.to(torch.bfloat16)  # noqa: E731
a = bf(torch.randn(m, d, generator=g)).cuda()
w1 = bf(torch.randn(hid, d, generator=g) / 16).cuda()
w2 = bf(torch.randn(d, hid, generator=g) / 45).cuda()
b1, b2 = torch.randn(hid, generator=g).cuda(), torch.randn(d, generator=g).cuda()
x = torch.randn(m, d, generator=g).cuda()
g1, be1 = torch.ones(d).cuda(), torch.zeros(d).cuda()
dy = bf(torch.randn(m, d, generator=g)).cuda()
g0 = torch.randn(m, d, generator=g).cuda()
pk = ops.ffn_pack_weights(w1, w2)
pt = ops.ffn_pack_weights(w2.t().contiguous(), w1.t().contiguous())
parts = torch.zeros(K.ffn_train_parts(m) * 512, device="cuda")
gk = K.ffn_train(a, pk, hid, b1, p, seed, 3, b2, x, 0.5, p, 4, ln1=(g1, be1), tape_derivative=True)[0]
names = ["start", "tile staged", "prologue done", "loop done", "drained", "tile free (stats done)", "reduced", "end"]
lib.ma_debug_ft_prof.argtypes = [ctypes.c_void_p]
for title, fn in (("forward", lambda: K.ffn_train(a, pk, hid, b1, p, seed, 3, b2, x, 0.5, p, 4, ln1=(g1, be1), tape_derivative=True)),
                  ("backward", lambda: K.ffn_train_bwd(dy, pt, hid, gk, x, g1, g0, parts, nxt=(0.5, p, seed, 9, None)))):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    acc, N = {}, 20
    for it in range(N):
        fn()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 48)()
        assert lib.ma_debug_ft_prof(buf) == 0
        t = np.array(buf[:], dtype=np.int64).reshape(3, 16)
        t0 = t[:, 0].min()
        for w in range(3):
            for k in range(8):
                acc.setdefault((w, k), []).append((t[w, k] - t0) / 100.0)
    print("%s: %-24s %14s %14s %14s   (us since the first of the three started; median of %d launches; +delta)" %
          (title, "phase", "wg 0", "wg 97", "wg 200", N))
    prev = [0, 0, 0]
    for k in range(8):
        med = [float(np.median(acc[(w, k)])) for w in range(3)]
        print("          %-24s " % names[k] + " ".join("%6.2f(+%5.2f)" % (med[w], med[w] - prev[w]) for w in range(3)))
        prev = med
