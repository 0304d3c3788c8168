"""CPU: the C-ABI library builds, loads, and exports every symbol include/mindaudio_amd.h declares.
No compute entry point is called (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = ""
    inc = os.path.join(ROOT, "include")
    for name in sorted(os.listdir(inc)):
        if name.endswith(".h"):
            text += open(os.path.join(inc, name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ma_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from mindaudio_amd import _build, _lib

    _build.build()
    return _lib.load()


def test_header_symbols_exported_and_bound(lib):
    from mindaudio_amd import _lib

    declared = _declared_symbols()
    assert len(declared) >= 10
    for sym in declared:
        assert hasattr(lib, sym), "library does not export %s" % sym
        assert sym in _lib.PROTOTYPES, "ctypes binding lacks %s" % sym
    assert sorted(_lib.PROTOTYPES) == declared


def test_host_only_entry_points(lib):
    from mindaudio_amd import _lib

    assert lib.ma_abi_version() == _lib.ABI_VERSION == 3
    assert lib.ma_num_frames(95984, 512, 128, 1) == 750  # tutorial shape (257, 750)
    assert lib.ma_num_frames(160000, 512, 160, 1) == 1001
    assert lib.ma_num_frames(160000, 512, 160, 0) == 997
    assert lib.ma_num_frames(300, 512, 160, 1) == -2  # n_fft > len -> ValueError in the mirror
    assert lib.ma_num_frames(1000, 512, 0, 1) == -3
    assert lib.ma_status_string(-3) == b"invalid hop_length"
    assert lib.ma_fbank_workspace_bytes(64, 1001) >= 64 * 32 * 8


def test_mirror_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np

    import mindaudio_amd
    from mindaudio_amd._lib import MindaudioAmdError

    with pytest.raises(MindaudioAmdError):
        mindaudio_amd.stft(np.zeros(1024, np.float32))
    with pytest.raises(MindaudioAmdError):
        mindaudio_amd.fbank(np.zeros((2, 1024), np.float32), n_fft=512)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mindaudio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), "%s mentions the oracle" % f


def test_host_tables_match_oracle():
    """Host-built tables (window, mel banks) of the product vs the oracle's restatement."""
    import numpy as np

    from mindaudio_amd import _host
    from oracle import speech_features as O

    assert np.array_equal(_host.centred_window_f64("hann", 400, 512), O._centered_window("hann", 400, 512))
    assert np.array_equal(_host.htk_fbanks_f64(257, 0.0, 8000.0, 80, 16000), O.melscale_fbanks(257, 0.0, 8000.0, 80, 16000))
    assert np.array_equal(_host.kaldi_banks_f64(80, 512, 16000.0, 20, 8000), O.kaldi_mel_banks(80, 512, 16000.0, 20, 8000)[0])
    with pytest.raises(ValueError):
        _host.centred_window_f64("hann", 600, 512)


def test_no_packed_fma_of_the_form_gfx950_miscomputes_beside_another_queue():
    """v_pk_fma_f32 vD, vA, vD, vC op_sel:[_,1,_] (destination = src1, low result from src1's high register) is exact alone and wrong
    in lanes 48-63 when MFMA-issuing waves of another kernel share the SIMD (tools/ubench/two_queue_pk.hip).  hipcc emits
    it for float2 code; the built library must not contain it (tools/check_pk_hazard.py disassembles every gfx950 code object)."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_pk_hazard", os.path.join(root, "tools", "check_pk_hazard.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    found, n_pk = mod.hazards(os.path.join(root, "mindaudio_amd", "lib", "libmindaudio_amd.so"))
    assert n_pk > 0, "the scanner found no packed FMA at all: is it still parsing the disassembly?"
    assert not found, found



def _load_tool(name):
    import importlib.util

    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_hazard_classifier_names_exactly_the_measured_form():
    """tools/pk_hazard_scan_external.py's line classifier: `hazard` = the one form tools/ubench/two_queue_pk.hip shows wrong (packed
    FMA, destination pair == src1 pair, src1 read hi -> lo); `wide` = any packed fp32 op whose destination aliases a source read with
    op_sel 1 (the members other than `hazard` were measured exact and are only listed)."""
    C = _load_tool("pk_hazard_scan_external").classify
    assert C("  v_pk_fma_f32 v[2:3], s[6:7], v[2:3], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,0,1] // 0001") == (True, True, True)
    assert C("  v_pk_fma_f32 v[2:3], v[8:9], v[2:3], v[6:7] op_sel_hi:[1,0,1]") == (True, False, False)      # no hi -> lo read
    assert C("  v_pk_fma_f32 v[2:3], v[2:3], v[8:9], v[6:7] op_sel:[1,0,0] op_sel_hi:[0,1,1]") == (True, False, True)  # dst == src0
    assert C("  v_pk_fma_f32 v[2:3], v[4:5], v[8:9], v[2:3] op_sel:[0,0,1] op_sel_hi:[1,1,0]") == (True, False, True)  # dst == src2
    assert C("  v_pk_fma_f32 v[2:3], v[4:5], v[8:9], v[6:7] op_sel:[0,1,0]") == (True, False, False)            # no aliasing
    assert C("  v_pk_add_f32 v[8:9], v[8:9], v[14:15] op_sel:[1,0] op_sel_hi:[0,1]") == (False, False, True)
    assert C("  v_pk_mul_f32 v[6:7], v[6:7], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]") == (False, False, True)
    assert C("  v_fma_f32 v2, v3, v2, v4") == (False, False, False)


def test_external_pk_hazard_scan_is_current_and_clean():
    """The packed-FMA hazard outside this library: at N = 8 our MFMA grids are the aggressor and RCCL's reduction
    kernels / the torch kernels of the step census the potential victims.  profiles/r06_pk_hazard_scan.txt is the committed result of
    tools/pk_hazard_scan_external.py over every gfx950 code object of torch's librccl.so and libtorch_hip.so; this test fails when (a)
    that scan found the hazardous form in a kernel on the watch list, or (b) the installed libraries are not the ones that were scanned
    (another torch / RCCL build: run the scan again).  MA_PK_SCAN_LIVE=1 re-runs the watch-list scan (minutes)."""
    path = os.path.join(ROOT, "profiles", "r06_pk_hazard_scan.txt")
    text = open(path).read()
    assert "verdict: CLEAN" in text, text[-400:]
    T = _load_tool("pk_hazard_scan_external")
    sizes = dict(re.findall(r"^library (\S+)\s+size (\d+) bytes", text, flags=re.M))
    assert set(sizes) == {"librccl.so", "libtorch_hip.so"}
    rccl = re.search(r"library librccl\.so.*?\n\s+all kernels:\s+v_pk_fma_f32 (\d+)\s+hazard (\d+)", text, flags=re.S)
    assert rccl and int(rccl.group(1)) > 0 and int(rccl.group(2)) == 0  # (it parsed RCCL's code, and RCCL is clean as a whole)
    try:
        libs = T.default_libs()
    except Exception:
        pytest.skip("torch not importable")
    for lib in libs:
        if not os.path.exists(lib):
            pytest.skip("%s not installed" % lib)
        assert os.path.getsize(lib) == int(sizes[os.path.basename(lib)]), \
            "%s is not the build that was scanned: run tools/pk_hazard_scan_external.py --out profiles/r06_pk_hazard_scan.txt" % lib
    if os.environ.get("MA_PK_SCAN_LIVE") == "1":
        for lib in libs:
            per_kernel, n_obj, n_kern = T.scan(lib, only_watched=True)
            assert n_kern > 0 and not [k for k, c in per_kernel.items() if c[1]]
