#!/usr/bin/env python3
"""Phase timeline of rows_packed_kernel<5, 3> (an input-gradient product with the LayerNorm backward in its epilogue) from a
-DMA_RP_PROF build (tools/lib_variant.sh rpprof "-DMA_RP_PROF" rows_packed.hip; MINDAUDIO_AMD_LIB=mindaudio_amd/lib/variants/rpprof.so):
wall_clock64 stamps (100 MHz) of wave 0 of three workgroups, cold operands (a 512 MiB fill between launches).
    python tools/rows_timeline.py [rows] [K ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mindaudio_amd import _lib, ops
from mindaudio_amd.train import kernels as K

lib = _lib.load()
lib.ma_debug_rp_prof.argtypes = [ctypes.c_void_p]
m = int(sys.argv[1]) if len(sys.argv) > 1 else 10200
names = ["entry", "weight prologue issued", "x landed", "LayerNorm statistics", "main loop done", "g landed", "tail issued", "stores retired"]
order = [0, 5, 6, 1, 2, 7, 3, 4]
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for k in [int(v) for v in sys.argv[2:]] or [512, 768]:
    g = torch.Generator(device="cuda").manual_seed(k)
    a = (torch.randn(m, k, device="cuda", generator=g) * 0.1).bfloat16()
    w = (torch.randn(256, k, device="cuda", generator=g) / 16).bfloat16()
    pk = ops.gemm_rows_pack(w)
    x = torch.randn(m, 256, device="cuda", generator=g)
    gamma = 1 + 0.1 * torch.randn(256, device="cuda", generator=g)
    gbuf = torch.randn(m, 256, device="cuda", generator=g)
    parts = torch.empty(K.rows_train_parts(m) * 512, device="cuda")
    fn = lambda: K.dense_lnbwd(a, pk, k, x, gamma, gbuf, parts, nxt=(0.5, 0.1, 7, 3, None))
    for _ in range(3): fn()
    acc = {}
    N = 20
    for it in range(N):
        flush.fill_(it)
        torch.cuda.synchronize()
        fn(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 24)()
        assert lib.ma_debug_rp_prof(buf) == 0
        t = np.array(buf[:], dtype=np.int64).reshape(3, 8)
        t0 = t[:, 0].min()
        for wg in range(3):
            for j in range(8):
                acc.setdefault((wg, j), []).append((t[wg, order[j]] - t0) / 100.0)
    print("M %d K %d   (us since the first start; median of %d; +delta)   wg 0 / wg 100 / last wg" % (m, k, N))
    prev = [0, 0, 0]
    for j in range(8):
        med = [float(np.median(acc[(wg, j)])) for wg in range(3)]
        print("  %-22s " % names[j] + " ".join("%7.2f(+%6.2f)" % (med[wg], med[wg] - prev[wg]) for wg in range(3)))
        prev = med
