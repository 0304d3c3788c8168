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

