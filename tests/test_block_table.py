"""The block launch table (include/mindaudio_amd.h: ma_block_table_*, ma_conformer_block_fwd_train / _bwd_train) on the host side:
the generated call list is current, and what the recorder stores for a call is what the call was made with.  No launches here - the
replayed step is compared bit for bit with the walked one in tests/test_train_step_gpu.py."""
import ctypes
import importlib.util
import os
import struct

import pytest

from mindaudio_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_call_list_is_current():
    spec = importlib.util.spec_from_file_location("gen_block_table", os.path.join(ROOT, "tools", "gen_block_table.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.main(check=True), "run python tools/gen_block_table.py after changing include/mindaudio_amd.h"


def test_replayable_entry_points_match_the_binding():
    lib = _lib.load()
    n = 0
    for name, (res, args) in _lib.PROTOTYPES.items():
        fid = lib.ma_block_table_entry_point(name.encode())
        if fid < 0:
            # every launch of the binding (int result, stream last ... but so do a few non-launches: not asserted the other way)
            continue
        n += 1
        # same parameter count as the ctypes prototype, a stream last, int result
        assert lib.ma_block_table_entry_point_params(fid) == len(args), name
        assert args[-1] is ctypes.c_void_p and res is ctypes.c_int, name
        seeds = lib.ma_block_table_entry_point_seeds(fid)
        for k, tp in enumerate(args):
            if (seeds >> k) & 1:
                assert tp is ctypes.c_uint32, (name, k)
    assert n >= 100
    for name in ("ma_ffn_train_bf16", "ma_ffn_train_bwd_bf16", "ma_gemm_k256_train_bf16", "ma_gemm_rows_train_bf16",
                 "ma_relpos_attention_train_bf16", "ma_relpos_attention_bwd_bf16", "ma_convmid_fwd_train", "ma_convmid_bwd_bn_bf16",
                 "ma_bn_swish_bwd_stage1_f32", "ma_reduce_splits_batch_f32", "ma_gemm_tn_direct_group_bf16",
                 "ma_layernorm_bwd_next_f32"):
        assert lib.ma_block_table_entry_point(name.encode()) >= 0, name
    # host pointer tables and the table's own entry points are not replayable; size queries are not launches
    for name, want in (("ma_fft_pow2_c32", -2), ("ma_conformer_block_fwd_train", -2), ("ma_gemm_tn_workspace_bytes", -1), ("nonsense", -1)):
        assert lib.ma_block_table_entry_point(name.encode()) == want, name


def test_recorder_stores_what_the_call_was_made_with():
    from mindaudio_amd.train.block_table import BlockTable

    lib = _lib.load()
    tab = BlockTable()
    fid = lib.ma_block_table_entry_point(b"ma_ffn_train_bwd_bf16")
    argtypes = _lib.PROTOTYPES["ma_ffn_train_bwd_bf16"][1]
    e = _lib.TrainEpilogue()
    e.mode, e.residual, e.ldr, e.alpha, e.p, e.seed, e.salt, e.ln_eps = 5, 0x7f00aa001000, 256, 0.5, 0.1, 1234, 55, 1e-5
    args = (ctypes.c_void_p(0x7f0000001000), 256, 10200, 2048, ctypes.c_void_p(0x7f0000002000), 0x7f0000003000, None, 2048,
            ctypes.c_void_p(0x7f0000004000), 256, ctypes.byref(e), None, ctypes.c_void_p(0xdead))
    tab.seed = 1234
    tab.segment(True, 7)
    tab._add("ma_ffn_train_bwd_bf16", fid, argtypes, lib.ma_block_table_entry_point_seeds(fid), args)
    h = tab.handle
    assert tab.calls(True, 7) == 1 and tab.calls(False, 7) == 0 and tab.calls(True, 6) == 0
    assert lib.ma_block_table_call_entry_point(h, 1, 7, 0) == fid and lib.ma_block_table_call_entry_point(h, 1, 7, 1) == -1
    words = [lib.ma_block_table_call_word(h, 1, 7, 0, k) for k in range(len(argtypes))]
    assert words[:10] == [0x7f0000001000, 256, 10200, 2048, 0x7f0000002000, 0x7f0000003000, 0, 2048, 0x7f0000004000, 256]
    assert words[10] == 0 and words[11] == -1  # the epilogue at offset 0 of the blob; chain = NULL
    size = lib.ma_block_table_call_blob(h, 1, 7, 0, None, 0)
    assert size == ctypes.sizeof(_lib.TrainEpilogue) and size % 8 == 0
    back = _lib.TrainEpilogue()
    lib.ma_block_table_call_blob(h, 1, 7, 0, ctypes.byref(back), size)
    assert bytes(back) == bytes(e)
    # an epilogue (or a seed argument) of another step's seed is refused: replay could not know which one to substitute
    tab.seed = 99
    with pytest.raises(_lib.MindaudioAmdError):
        tab._add("ma_ffn_train_bwd_bf16", fid, argtypes, lib.ma_block_table_entry_point_seeds(fid), args)
    # floats travel as the bits of a double, seeds are checked, host arrays are copied whole
    fid2 = lib.ma_block_table_entry_point(b"ma_dropout_bwd_bf16")
    at2 = _lib.PROTOTYPES["ma_dropout_bwd_bf16"][1]
    a2 = (1, 2, 3, 4, 5, 6, 0.5, None, 0.1, 99, 7, None)
    tab.segment(False, 0)
    tab._add("ma_dropout_bwd_bf16", fid2, at2, lib.ma_block_table_entry_point_seeds(fid2), a2)
    w = [lib.ma_block_table_call_word(h, 0, 0, 0, k) for k in range(len(at2))]
    assert struct.unpack("<d", struct.pack("<q", w[6]))[0] == 0.5 and struct.unpack("<d", struct.pack("<q", w[8]))[0] == 0.1
    assert w[9] == 99 and w[10] == 7
    with pytest.raises(_lib.MindaudioAmdError):
        tab._add("ma_dropout_bwd_bf16", fid2, at2, lib.ma_block_table_entry_point_seeds(fid2), a2[:9] + (98,) + a2[10:])
    items = (_lib.TnDirectItem * 3)()
    items[2].Mo = 768
    fid3 = lib.ma_block_table_entry_point(b"ma_gemm_tn_direct_group_bf16")
    tab._add("ma_gemm_tn_direct_group_bf16", fid3, _lib.PROTOTYPES["ma_gemm_tn_direct_group_bf16"][1], 0, (items, 3, None))
    assert lib.ma_block_table_call_blob(h, 0, 0, 1, None, 0) == 3 * ctypes.sizeof(_lib.TnDirectItem)
    # a block without entries is an error, not an empty success
    assert lib.ma_conformer_block_fwd_train(h, 5, 1, None) == _lib.MA_ERR_INVALID_ARG
    # wrong word count / unknown entry point / unaligned blob
    w17 = (ctypes.c_int64 * 17)()
    assert lib.ma_block_table_add(h, 0, 1, fid, w17, 17, None, 0) == _lib.MA_ERR_INVALID_ARG
    assert lib.ma_block_table_add(h, 0, 1, 10 ** 6, w17, 17, None, 0) == _lib.MA_ERR_INVALID_ARG
    assert lib.ma_block_table_add(h, 0, 99, fid, w17, 13, None, 0) == _lib.MA_ERR_INVALID_ARG
    assert lib.ma_block_table_add(h, 0, 1, fid, w17, 13, b"abc", 3) == _lib.MA_ERR_INVALID_ARG


def test_recording_is_per_thread():
    """While one thread fills a table, `_lib.load()` hands the recording proxy to THAT thread only."""
    import threading

    from mindaudio_amd.train.block_table import BlockTable

    tab, seen = BlockTable(), {}
    with tab.recording(5):
        seen["recorder"] = type(_lib.load()).__name__
        th = threading.Thread(target=lambda: seen.__setitem__("other", type(_lib.load()).__name__))
        th.start()
        th.join()
        with pytest.raises(_lib.MindaudioAmdError):
            tab.recording(6).__enter__()
    seen["after"] = type(_lib.load()).__name__
    assert seen == {"recorder": "_RecordingLib", "other": "CDLL", "after": "CDLL"}


def test_a_launch_the_table_cannot_reissue_marks_the_table_and_is_still_issued():
    """A call the table cannot replay inside a block does not stop the walked step it is met in (the step is complete and correct
    all the same): the launch is issued, the table is marked `broken`, records nothing further, and the engine walks that batch
    shape from then on (ConformerCTCTrainStep._table_recorded)."""
    from mindaudio_amd.train.block_table import BlockTable

    tab = BlockTable()
    with tab.recording(1):
        lib = _lib.load()
        tab.segment(False, 0)
        assert tab.broken is None
        rc = lib.ma_fft_pow2_c32(None, None, 0, 0, 0, None, None)  # issued: the library itself answers (invalid arguments)
        assert rc == _lib.MA_ERR_INVALID_ARG and tab.broken is not None and "ma_fft_pow2_c32" in tab.broken
        # size queries pass through, recorded or not
        assert lib.ma_gemm_tn_workspace_bytes(256, 256, 1024) > 0
        tab.segment(None, 0)
    assert tab.calls(False, 0) == 0


def test_a_seed_mismatch_while_recording_marks_the_table_instead_of_raising():
    from mindaudio_amd.train.block_table import BlockTable

    tab = BlockTable()
    calls = []
    with tab.recording(5):
        tab.segment(True, 0)
        name = next(n for n in _lib.PROTOTYPES if tab.lib.ma_block_table_entry_point(n.encode()) >= 0
                    and tab.lib.ma_block_table_entry_point_seeds(tab.lib.ma_block_table_entry_point(n.encode())))
        fid = tab.lib.ma_block_table_entry_point(name.encode())
        mask = tab.lib.ma_block_table_entry_point_seeds(fid)
        argtypes = _lib.PROTOTYPES[name][1]
        args = [None if t is __import__("ctypes").c_void_p else 0 for t in argtypes]
        k = next(i for i in range(len(argtypes)) if (mask >> i) & 1)
        args[k] = 6  # a seed that is not the step's
        wrapped = tab._wrap(name, lambda *a: calls.append(a) or 0)
        assert wrapped(*args) == 0 and len(calls) == 1  # issued
        assert tab.broken is not None and "seed" in tab.broken and tab.recorded == 0
        tab.segment(None, 0)


@pytest.mark.parametrize("dec_pending", [False, True])
@pytest.mark.parametrize("blocks,group", [(12, 6), (12, 5), (2, 6), (7, 3), (1, 1)])
def test_replayed_backward_launches_the_gradient_buckets_where_the_walked_one_does(monkeypatch, blocks, group, dec_pending):
    """N > 1: a finished group's gradient buckets go on the wire (reducer.launch) behind its direct weight-gradient products.  The
    replayed backward pass (one C call per block) must issue the same spans at the same points as _layer_done / _flush_direct do
    when the blocks are walked - checked here on the host logic alone (no launches: the library and the table are stand-ins).
    dec_pending (round 6): the decoder's long-contraction weight gradients ride in the encoder's FIRST direct group, so the
    decoder's bucket must leave behind that group - once, in the walked and in the replayed pass alike."""
    import ctypes
    from types import SimpleNamespace

    from mindaudio_amd import _host
    from mindaudio_amd.train import engine as E

    def make():
        eng = object.__new__(E.ConformerCTCTrainStep)
        log = []
        eng.L, eng.d, eng.hidden, eng.K, eng.p_drop = blocks, 256, 2048, None, 0.1
        eng.dw_group_blocks, eng._dw_direct, eng._wg, eng._dq, eng._dq_blocks = group, True, None, None, []
        eng.layer_names = [["l%d.first" % i, "l%d.last" % i] for i in range(blocks)]
        eng.fp = SimpleNamespace(span=lambda names: (names[0], names[-1]))
        eng.reducer = SimpleNamespace(launch=lambda lo, hi: log.append(("bucket", lo, hi)))
        item = SimpleNamespace(data_ptr=lambda: 0)
        eng._dw_cur = dict(layers=[(item, item, 1)] * blocks)
        eng._dec_bucket_pending, eng.dec_names = dec_pending, ["dec.first", "dec.last"]
        return eng, log

    prev = _host.swap_pinned(ctypes.c_void_p(0))
    try:
        # walked: what _layer_done does after each block's launches
        walked, wlog = make()
        stub = SimpleNamespace(ma_reduce_splits_batch_f32=lambda *a: wlog.append(("block", a[2])) or 0)
        monkeypatch.setattr(E._lib, "load", lambda: stub)
        for li in reversed(range(blocks)):
            wlog.append(("launches", li))
            walked._layer_done(li)
        # replayed: the table stands in for the block's launches
        replayed, rlog = make()
        table = SimpleNamespace(backward=lambda li, seed, stream: rlog.append(("launches", li)))
        ctx = dict(seed=1, b=1, t2=1, mask_rows=None, att_mask=None, pos_all=None, table=dict(state="replay", table=table))
        replayed._blocks_backward_fused(None, None, None, ctx)
    finally:
        _host.swap_pinned(prev)
    assert [e for e in wlog if e[0] != "block"] == rlog
    assert sum(1 for e in rlog if e[0] == "bucket") == blocks + int(dec_pending) and walked._dq_blocks == replayed._dq_blocks == []
    if dec_pending:
        dec = [i for i, e in enumerate(rlog) if e == ("bucket", "dec.first", "dec.last")]
        first_group = min(group, blocks)
        launches = [i for i, e in enumerate(rlog) if e[0] == "launches"]
        # exactly once, behind the first group's last block and before the next block's launches
        assert len(dec) == 1 and dec[0] > launches[first_group - 1] and (len(launches) == first_group or dec[0] < launches[first_group])
        assert not walked._dec_bucket_pending and not replayed._dec_bucket_pending


def test_replay_dispatches_typed_calls_and_reports_the_failing_entry():
    """No GPU needed: entries whose arguments the entry points reject (NULL buffers) make the replay stop at that entry with the
    entry point's own status - which shows that the generated typed call reached the right function with the recorded words."""
    from mindaudio_amd.train.block_table import BlockTable

    lib = _lib.load()
    tab = BlockTable()
    tab.seed = 5
    tab.segment(False, 3)
    for name, args in (("ma_cast_f32_bf16", (None, None, 16, None)),
                       ("ma_dropout_bwd_bf16", (None, 0, None, 0, 4, 4, 1.0, None, 0.1, 5, 6, None))):
        fid = lib.ma_block_table_entry_point(name.encode())
        tab._add(name, fid, _lib.PROTOTYPES[name][1], lib.ma_block_table_entry_point_seeds(fid), args)
    assert tab.calls(False, 3) == 2
    assert lib.ma_conformer_block_fwd_train(tab.handle, 3, 7, None) == lib.ma_cast_f32_bf16(None, None, 16, None) == _lib.MA_ERR_INVALID_ARG
    assert lib.ma_block_table_failed_call(tab.handle) == 0
    with pytest.raises(ValueError):
        tab.forward(3, 7, None)
    assert lib.ma_conformer_block_bwd_train(tab.handle, 3, 7, None) == _lib.MA_ERR_INVALID_ARG  # nothing recorded that way


def test_block_table_from_plain_c(tmp_path):
    """tests/c/block_table_abi.c: the table filled, read back and replayed from C99 through include/mindaudio_amd.h alone."""
    import shutil
    import subprocess

    from mindaudio_amd import _build

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    lib_dir = os.path.dirname(_build.build())
    exe = str(tmp_path / "block_table_abi")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "block_table_abi.c"), "-o", exe, "-L", lib_dir, "-lmindaudio_amd",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert out.returncode == 0 and out.stdout.decode().strip().endswith("ok"), out.stdout.decode()
