#!/usr/bin/env python3
"""Per-phase cycle breakdown of feat512_kernel from an -DMA_PROFILE build (tools only)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
prof_lib = os.path.join(ROOT, "mindaudio_amd", "lib", "libmindaudio_amd_prof.so")
src = os.path.join(ROOT, "mindaudio_amd", "csrc", "features.hip")
if not os.path.exists(prof_lib) or os.path.getmtime(prof_lib) < os.path.getmtime(src) or "--rebuild" in sys.argv:
    # every source goes in (the binding resolves all symbols of the header); only features.hip looks at MA_PROFILE
    import glob
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-DMA_PROFILE", "-shared"]
                          + sorted(glob.glob(os.path.join(os.path.dirname(src), "*.hip"))) + ["-o", prof_lib])
os.environ["MINDAUDIO_AMD_LIB"] = prof_lib
import numpy as np, torch
from mindaudio_amd import _host, _lib
lib = _lib.load()
B, N, T = 64, 160000, 1001
x = torch.from_numpy((0.1*np.random.RandomState(1234).randn(B, N)).astype(np.float32)).cuda()
win = _host.device_window("hann", 512, 512, x.device)
bank = _host.device_htk_bank(512, 0.0, 8000.0, 80, 16000, x.device)
ws = _host.workspace(lib.ma_fbank_workspace_bytes(B, T), x.device)
out = torch.empty((B, 80, T), device="cuda")
prof = torch.zeros(64, dtype=torch.int64, device="cuda")
lib.ma_debug_set_prof.argtypes = [ctypes.c_void_p]
lib.ma_debug_set_prof(ctypes.c_void_p(prof.data_ptr()))
st = _host.current_stream_ptr()
KALDI = "--kaldi" in sys.argv  # the Kaldi front end of the Conformer loader (25 ms frames, pre-emphasis, scalar mean) instead
if KALDI:
    kwin = _host.device_kaldi_window(400, x.device)
    kbank = _host.device_kaldi_bank(80, 512, 16000.0, 20.0, 8000.0, x.device)
    klens = torch.full((B,), N, dtype=torch.int64, device="cuda")
    kT = (N - 400) // 160 + 1
    kout = torch.empty((B, kT, 80), device="cuda")
    kframes = torch.empty((B,), dtype=torch.int64, device="cuda")
    kws = _host.workspace(lib.ma_fbank_workspace_bytes(B, kT), x.device)
def run():
    if KALDI:
        assert lib.ma_fbank_kaldi_f32(_host.ptr(x), _host.ptr(klens), B, N, N, 400, 160, 512, _host.ptr(kwin), kbank.ref(), 0.97,
                                      _host.ptr(kout), _host.ptr(kframes), _host.ptr(kws), kws.numel(), st) == 0
        return
    assert lib.ma_fbank_db_f32(_host.ptr(x), B, N, N, 512, 160, _host.ptr(win), 1, 1, bank.ref(), 2.0, 10.0, 1e-10, 0.0, -1.0, _host.ptr(out), _host.ptr(ws), ws.numel(), st) == 0
for _ in range(3): run()
torch.cuda.synchronize(); prof.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); n = 10
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
p = prof.cpu().numpy().astype(float) / n
# ---- timeline of block 0 / wave 0 (wall_clock64 = 100 MHz ticks) ----
torch.cuda.synchronize(); prof.zero_(); run(); torch.cuda.synchronize()
stamps = prof.cpu().numpy()[16:56]
stamps = [(int(v) >> 56, int(v) & ((1 << 56) - 1)) for v in stamps if v != 0]
names_s = {1: "entry", 2: "tables ready", 3: "samples consumed", 4: "fft done", 5: "mel done", 6: "end"}
if stamps:
    t0 = stamps[0][1]
    print("timeline (us since entry):", ", ".join("%s %.2f" % (names_s.get(k, k), (t - t0) / 100.0) for k, t in stamps))
lib.ma_debug_set_prof(ctypes.c_void_p(0))
def timeit(flags, reps=30):
    lib.ma_debug_set_flags(flags)
    for _ in range(3): run()
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for flags, nm in ((0, "full"), (1, "no sample loads"), (2, "mel loop 1 iter"), (4, "no out stores"), (8, "no fft"), (6, "no mel loop+no stores"), (14, "no fft, no mel, no stores"), (15, "all off"), (16, "launch + tables + first staging only"), (17, "launch + tables only"), (32, "launch only")):
    print("ablation %-28s %7.1f us" % (nm, timeit(flags)))
names = ["loop/tail", "loads+window", "fft+split+P", "barrier1", "mel+log+store", "reduce/barriers"]
tot = p[:6].sum()
print("kernel %.1f us (instrumented)" % (e0.elapsed_time(e1) / n * 1e3))
for i, nm in enumerate(names):
    print("%-16s %12.0f wave-cycles  %5.1f%%" % (nm, p[i], 100 * p[i] / tot))
