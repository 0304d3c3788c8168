#!/usr/bin/env python3
"""Phase timeline of gemm_bf16_8ph_kernel from a -DMA_G8_PROF build (tools/lib_variant.sh g8prof "-DMA_G8_PROF" gemm_bf16.hip;
MINDAUDIO_AMD_LIB=mindaudio_amd/lib/variants/g8prof.so): wall_clock64 stamps (100 MHz) of wave 0 of three workgroups."""
import ctypes, os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mindaudio_amd import _lib, ops
lib = _lib.load()
lib.ma_debug_g8_prof.argtypes = [ctypes.c_void_p]
names = ["entry", "first K-tile issued", "first K-tile landed", "main loop done", "epilogue issued", "stores retired"]
shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(4096, 4096, 1024), (4096, 4096, 4096), (16384, 4096, 1024)]
for (m, n, k) in shapes:  # (argv: MxNxK ...)
    a = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") / math.sqrt(k)).bfloat16()
    o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.gemm(a, w, out=o)
    for _ in range(5): fn()
    acc = {}
    N = 20
    for it in range(N):
        fn(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 24)()
        assert lib.ma_debug_g8_prof(buf) == 0
        t = np.array(buf[:], dtype=np.int64).reshape(3, 8)
        t0 = t[:, 0].min()
        for wg in range(3):
            for j in range(6):
                acc.setdefault((wg, j), []).append((t[wg, j] - t0) / 100.0)
    print("M %d N %d K %d   (us since the first start; median of %d; +delta)   wg 0 / wg 100 / last wg" % (m, n, k, N))
    prev = [0, 0, 0]
    for j in range(6):
        med = [float(np.median(acc[(wg, j)])) for wg in range(3)]
        print("  %-22s " % names[j] + " ".join("%7.2f(+%6.2f)" % (med[wg], med[wg] - prev[wg]) for wg in range(3)))
        prev = med
