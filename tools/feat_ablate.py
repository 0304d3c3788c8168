#!/usr/bin/env python3
"""Ablation timings of feat512_kernel on the PRODUCTION instruction stream: a -DMA_ABLATE build of the library (no timing code)
whose kernel skips parts according to the bits of MA_FEAT_DBG: 1 no sample staging, 2 one mel step per row, 4 no output stores,
8 no FFT, 16 return after the tables, 32 return at entry, 64 no unit_min store.  One child process per setting (the flag is read once)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "mindaudio_amd", "lib", "libmindaudio_amd_ablate.so")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    os.environ["MINDAUDIO_AMD_LIB"] = lib
    import numpy as np, torch
    from mindaudio_amd import _host, _lib
    l = _lib.load()
    B, N, T = int(os.environ.get("B", 64)), 160000, 1001
    x = torch.from_numpy((0.1 * np.random.RandomState(1234).randn(B, N)).astype(np.float32)).cuda()
    win = _host.device_window("hann", 512, 512, x.device)
    bank = _host.device_htk_bank(512, 0.0, 8000.0, 80, 16000, x.device)
    ws = _host.workspace(l.ma_fbank_workspace_bytes(B, T), x.device)
    out = torch.empty((B, 80, T), device="cuda")
    st = _host.current_stream_ptr()
    def run():
        assert l.ma_fbank_db_f32(_host.ptr(x), B, N, N, 512, 160, _host.ptr(win), 1, 1, bank.ref(), 2.0, 10.0, 1e-10, 0.0, -1.0, _host.ptr(out), _host.ptr(ws), ws.numel(), st) == 0
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print("%.1f" % (e0.elapsed_time(e1) / 50 * 1e3))
    sys.exit(0)
src = sorted(glob.glob(os.path.join(ROOT, "mindaudio_amd", "csrc", "*.hip")))
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-DMA_ABLATE", "-shared"] + src + ["-o", lib],
                      stderr=subprocess.DEVNULL)
names = {0: "full", 1: "no staging", 2: "mel 1 step/row", 4: "no out stores", 68: "no out stores, no unit_min", 8: "no fft", 6: "mel 1 step + no stores",
         12: "no fft, no stores", 14: "no fft, mel 1 step, no stores", 79: "all off", 16: "launch + tables + first staging", 32: "launch only"}
for cfg in ("4x3",):
    for flags, nm in names.items():
        env = dict(os.environ, MA_FEAT_DBG=str(flags))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, stdout=subprocess.PIPE, text=True)
        print("cfg %s  %-34s %s us" % (cfg, nm, r.stdout.strip()))
