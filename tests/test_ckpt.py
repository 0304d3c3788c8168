"""MindSpore .ckpt importer (mindaudio_amd/utils/ckpt.py).  Parity unpinned: MindSpore is not installed and no
reference checkpoint exists in the container, so the wire format is exercised through the package's own writer, against
google.protobuf under MindSpore's published schema (both directions), and the name mapping against names derived from the
reference's cell attributes."""
import numpy as np
import pytest

from mindaudio_amd.utils import ckpt as K


def _to_reference_names(state):
    """What a MindSpore training run would have saved: the product's inverse name map (utils.ckpt.to_reference_names) under the
    train network's `network.` prefix, plus optimizer / loss-scale state the importer has to drop."""
    out = K.to_reference_names(state, prefix="network.")
    out["global_step"] = np.array([123], np.int32)
    out["moments.network.encoder.after_norm.gamma"] = np.zeros(256, np.float32)
    out["scale_sense"] = np.array(1024.0, np.float32)
    return out


def test_wire_format_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    params = {"a.weight": rng.randn(7, 5).astype(np.float32), "b": rng.randn(3).astype(np.float16),
              "step": np.array([5], np.int32), "scalar": np.array(2.5, np.float32),
              "big": rng.randn(1000, 16).astype(np.float32), "idx": np.arange(10, dtype=np.int64) - 5}
    path = str(tmp_path / "x.ckpt")
    K.write_mindspore_ckpt(path, params, slice_bytes=20000)  # "big" is written as 4 slices with the same tag
    got = K.read_mindspore_ckpt(path)
    assert set(got) == set(params)
    for k, v in params.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(got[k], v), k
    raw = open(path, "rb").read()
    assert raw.count(b"big") == 4 and raw[0] == 0x0A  # field 1, length-delimited
    open(path, "wb").write(raw[:-7])
    with pytest.raises(ValueError):
        K.read_mindspore_ckpt(path)


def test_import_into_asr_model(tmp_path):
    import torch

    from mindaudio_amd.conformer.asr_model import create_asr_model

    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=2)
    dec = dict(attention_heads=4, linear_units=2048, num_blocks=1)
    torch.manual_seed(1)
    src = create_asr_model(80, 100, conf, None, ctc_weight=0.3, decoder_conf=dec)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.01)
        bn = src.encoder.encoders[0].conv_module.norm
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    ref_named = _to_reference_names(src.state_dict())
    assert "network.encoder.embed.conv.2.conv2d.weight" in ref_named
    assert ref_named["network.encoder.encoders.0.conv_module.depthwise_conv.conv1d.weight"].shape == (256, 1, 1, 15)
    assert "network.encoder.encoders.1.self_attn.linear_pos.dense.weight" in ref_named
    path = str(tmp_path / "avg_30.ckpt")
    K.write_mindspore_ckpt(path, ref_named)
    torch.manual_seed(2)
    dst = create_asr_model(80, 100, conf, None, ctc_weight=0.3, decoder_conf=dec)
    missing, unexpected = K.load_mindspore_checkpoint(dst, path)
    assert not missing and not unexpected
    a, b = src.state_dict(), dst.state_dict()
    for k in a:
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(a[k], b[k]), k
    # strictness: a parameter the module does not have / a missing one / a wrong shape
    bad = dict(ref_named)
    bad["network.encoder.extra.weight"] = np.zeros(3, np.float32)
    K.write_mindspore_ckpt(path, bad)
    with pytest.raises(KeyError):
        K.load_mindspore_checkpoint(dst, path)
    assert K.load_mindspore_checkpoint(dst, path, strict=False)[1] == ["encoder.extra.weight"]
    bad = dict(ref_named)
    bad["network.ctc.ctc_lo.weight"] = np.zeros((99, 256), np.float32)
    K.write_mindspore_ckpt(path, bad)
    with pytest.raises(ValueError):
        K.load_mindspore_checkpoint(dst, path)


def test_strict_import_with_global_cmvn_and_stale_caches(tmp_path):
    """The shipped conformer.yaml sets cmvn_file: GlobalCMVN's mean/istd are constructor data in the reference (layers/cmvn.py),
    never in a .ckpt, so a strict load must not ask for them; and a load drops every bf16 / packed weight copy made earlier."""
    import torch

    from mindaudio_amd.conformer.asr_model import create_asr_model

    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=1)
    cmvn = (np.linspace(-1, 1, 80), np.linspace(0.5, 2, 80))
    torch.manual_seed(3)
    src = create_asr_model(80, 50, conf, cmvn)
    assert not [k for k in src.state_dict() if "cmvn" in k]
    path = str(tmp_path / "cmvn.ckpt")
    K.write_mindspore_ckpt(path, _to_reference_names(src.state_dict()))
    torch.manual_seed(4)
    dst = create_asr_model(80, 50, conf, cmvn)
    dst.encoder._prepared = {"stale": True}
    dst.ctc._w = ("stale",)
    missing, unexpected = K.load_mindspore_checkpoint(dst, path)  # strict
    assert not missing and not unexpected
    assert dst.encoder._prepared is None and dst.ctc._w is None
    assert torch.equal(dst.encoder.cmvn_istd, torch.as_tensor(cmvn[1], dtype=torch.float32))
    for k, v in src.state_dict().items():
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(v, dst.state_dict()[k]), k
    # torch's own load_state_dict drops them too
    dst.encoder._prepared = {"stale": True}
    dst.ctc._w = ("stale",)
    dst.load_state_dict(src.state_dict())
    assert dst.encoder._prepared is None and dst.ctc._w is None


def _checkpoint_message_class():
    """MindSpore's checkpoint schema (mindspore/ccsrc/utils/checkpoint.proto as published with 2.3: Checkpoint { repeated Value value = 1 },
    Value { required string tag = 1; required TensorProto tensor = 2 }, TensorProto { repeated int64 dims = 1; required string
    tensor_type = 2; required bytes tensor_content = 3 }) built as a dynamic google.protobuf message: an independent implementation of
    the wire format to hold the hand-written reader / writer against."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.name, fdp.syntax = "ms_checkpoint_for_test.proto", "proto2"
    t = fdp.message_type.add()
    t.name = "TensorProto"
    for name, number, label, typ in (("dims", 1, "LABEL_REPEATED", "TYPE_INT64"), ("tensor_type", 2, "LABEL_REQUIRED", "TYPE_STRING"),
                                     ("tensor_content", 3, "LABEL_REQUIRED", "TYPE_BYTES")):
        f = t.field.add()
        f.name, f.number, f.label, f.type = name, number, getattr(f, label), getattr(f, typ)
    c = fdp.message_type.add()
    c.name = "Checkpoint"
    v = c.nested_type.add()
    v.name = "Value"
    f = v.field.add()
    f.name, f.number, f.label, f.type = "tag", 1, f.LABEL_REQUIRED, f.TYPE_STRING
    f = v.field.add()
    f.name, f.number, f.label, f.type, f.type_name = "tensor", 2, f.LABEL_REQUIRED, f.TYPE_MESSAGE, ".TensorProto"
    f = c.field.add()
    f.name, f.number, f.label, f.type, f.type_name = "value", 1, f.LABEL_REPEATED, f.TYPE_MESSAGE, ".Checkpoint.Value"
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    desc = pool.FindMessageTypeByName("Checkpoint")
    if hasattr(message_factory, "GetMessageClass"):
        return message_factory.GetMessageClass(desc)
    return message_factory.MessageFactory(pool).GetPrototype(desc)


def test_wire_format_against_google_protobuf(tmp_path):
    """Second anchor for the wire format (the first is the package's own round trip): files of write_mindspore_ckpt parse with
    google.protobuf under MindSpore's schema - tags, dims, type names, bytes - and a file serialised BY google.protobuf (unpacked and
    packed dims, a negative dim, a sliced parameter) reads back through read_mindspore_ckpt."""
    pytest.importorskip("google.protobuf")
    Ck = _checkpoint_message_class()
    rng = np.random.RandomState(3)
    params = {"network.encoder.after_norm.gamma": rng.randn(256).astype(np.float32), "w": rng.randn(4, 3, 1, 5).astype(np.float16),
              "epoch_num": np.array([7], np.int32), "scale_sense": np.array(1024.0, np.float32),
              "big": rng.randn(300, 8).astype(np.float32), "i64": np.arange(6, dtype=np.int64).reshape(2, 3) - 3}
    path = str(tmp_path / "ours.ckpt")
    K.write_mindspore_ckpt(path, params, slice_bytes=4000)
    msg = Ck()
    msg.ParseFromString(open(path, "rb").read())
    assert msg.IsInitialized()
    seen = {}
    for val in msg.value:
        seen.setdefault(val.tag, []).append(val.tensor)
    assert set(seen) == set(params) and len(seen["big"]) == 3 and all(len(v) == 1 for k, v in seen.items() if k != "big")
    names = {"float32": "Float32", "float16": "Float16", "int32": "Int32", "int64": "Int64"}
    for k, arr in params.items():
        assert all(t.tensor_type == names[arr.dtype.name] for t in seen[k]), k
        data = b"".join(t.tensor_content for t in seen[k])
        assert data == arr.tobytes(), k
        if k == "big":
            assert [list(t.dims) for t in seen[k]] == [[100, 8]] * 3
        else:
            assert list(seen[k][0].dims) == (list(arr.shape) or [0]), k
    # the other direction: google.protobuf writes, the package reads
    msg = Ck()
    for k, arr in params.items():
        pieces = np.array_split(arr, 3) if k == "big" else [arr]
        for piece in pieces:
            val = msg.value.add()
            val.tag = k
            val.tensor.dims.extend(piece.shape or (0,))
            val.tensor.tensor_type = names[arr.dtype.name]
            val.tensor.tensor_content = piece.tobytes()
    path2 = str(tmp_path / "theirs.ckpt")
    open(path2, "wb").write(msg.SerializeToString())
    got = K.read_mindspore_ckpt(path2)
    assert set(got) == set(params)
    for k, arr in params.items():
        assert got[k].dtype == arr.dtype and got[k].shape == arr.shape and np.array_equal(got[k], arr), k
