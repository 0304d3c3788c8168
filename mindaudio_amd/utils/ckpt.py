"""MindSpore `.ckpt` -> this package's modules: the importer that lets the released Conformer checkpoint
(examples/conformer/readme.md:118-136) be evaluated here; examples/conformer/predict.py:66-67 does
`load_param_into_net(network, load_checkpoint(path))`.

The checkpoint format belongs to MindSpore 2.3.0 (requirements.txt:1), which is not vendored and not installed here, so this
file restates its published layout — **parity unpinned** (no reference checkpoint can be produced or fetched in this
container; the tests round-trip through the writer below):

    message Checkpoint { message Value { required string tag = 1; required TensorProto tensor = 2; } repeated Value value = 1; }
    message TensorProto { repeated int64 dims = 1; required string tensor_type = 2; required bytes tensor_content = 3; }

`save_checkpoint` writes one serialised Checkpoint per parameter slice back to back (a protobuf stream concatenates), large
parameters split along axis 0 into several values with the same tag.  Parsed here with a 40-line wire-format reader: no protobuf
schema compilation, no MindSpore.

Parameter names follow the reference's cell attributes (mindaudio/models/conformer.py, mindaudio/models/layers/*.py):
wrappers `Dense.dense`, `Conv1d.conv1d`, `Conv2d.conv2d`, `nn.SequentialCell` indices, BatchNorm `gamma/beta/moving_*`,
`nn.Embedding.embedding_table`; `convert_names` maps them onto the names of mindaudio_amd.conformer.asr_model.ASRModel.
"""
import re

import numpy as np

__all__ = ["read_mindspore_ckpt", "write_mindspore_ckpt", "convert_names", "to_reference_names", "read_epoch_num",
           "load_mindspore_checkpoint"]

_DTYPES = {"Float32": np.float32, "Float16": np.float16, "Float64": np.float64, "Int32": np.int32, "Int64": np.int64,
           "Int16": np.int16, "Int8": np.int8, "UInt8": np.uint8, "UInt16": np.uint16, "UInt32": np.uint32,
           "UInt64": np.uint64, "Bool": np.bool_}
_NAMES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


def _varint(buf, pos):
    val, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of one message; value = int (varint / fixed) or memoryview (length-delimited)."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 2:
            n, pos = _varint(buf, pos)
            val = buf[pos:pos + n]
            if len(val) != n:
                raise ValueError("truncated checkpoint")
            pos += n
        elif wt == 1:
            val = int.from_bytes(buf[pos:pos + 8], "little")
            pos += 8
        elif wt == 5:
            val = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d (encrypted or corrupt checkpoint?)" % wt)
        yield num, wt, val


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def read_mindspore_ckpt(path):
    """{parameter name: ndarray} of a MindSpore checkpoint file, slices of one parameter joined along axis 0."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    parts = {}
    for num, wt, val in _fields(buf):
        if num != 1 or wt != 2:
            continue
        tag, tensor = None, None
        for n2, w2, v2 in _fields(val):
            if n2 == 1 and w2 == 2:
                tag = bytes(v2).decode("utf-8")
            elif n2 == 2 and w2 == 2:
                tensor = v2
        if tag is None or tensor is None:
            raise ValueError("checkpoint value without tag / tensor")
        dims, ttype, content = [], None, b""
        for n3, w3, v3 in _fields(tensor):
            if n3 == 1 and w3 == 0:
                dims.append(_signed(v3))
            elif n3 == 1 and w3 == 2:  # packed encoding
                p = 0
                while p < len(v3):
                    d, p = _varint(v3, p)
                    dims.append(_signed(d))
            elif n3 == 2 and w3 == 2:
                ttype = bytes(v3).decode("utf-8")
            elif n3 == 3 and w3 == 2:
                content = v3
        if ttype == "BFloat16":
            arr = (np.frombuffer(content, np.uint16).astype(np.uint32) << 16).view(np.float32)
        elif ttype in _DTYPES:
            arr = np.frombuffer(content, _DTYPES[ttype])
        else:
            raise ValueError("unsupported tensor_type %r for %s" % (ttype, tag))
        shape = tuple(dims) if dims and dims != [0] else ()
        parts.setdefault(tag, []).append(arr.reshape(shape) if shape else arr.reshape(()) if arr.size == 1 else arr)
    return {k: (v[0] if len(v) == 1 else np.concatenate(v, axis=0)).copy() for k, v in parts.items()}


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(num, payload):
    return _enc_varint(num << 3 | 2) + _enc_varint(len(payload)) + payload


def write_mindspore_ckpt(path, params, slice_bytes=512 * 1024 * 1024):
    """Serialise {name: ndarray} in the layout above (the exporter of `ConformerCTCTrainStep.sync_to_module()` weights, and
    what the importer's tests read back)."""
    with open(path, "wb") as f:
        for name, arr in params.items():
            arr = np.asarray(arr, order="C")  # (ascontiguousarray would turn a scalar into shape (1,))
            n_slices = max(1, -(-arr.nbytes // slice_bytes)) if arr.ndim else 1
            for piece in (np.array_split(arr, n_slices) if n_slices > 1 else [arr]):
                dims = b"".join(_enc_varint(1 << 3 | 0) + _enc_varint(d) for d in (piece.shape or (0,)))
                tensor = dims + _ld(2, _NAMES[piece.dtype.name].encode()) + _ld(3, piece.tobytes())
                f.write(_ld(1, _ld(1, name.encode()) + _ld(2, tensor)))


_SKIP = re.compile(r"^(moments?\d*\.|moment[12]\.|adam_[mv]\.|accu_grads\.|global_step$|learning_rate$|scale_sense$|step$|"
                   r"epoch_num$|step_num$|loss_scale$|cur_step$|beta[12]_power$|current_iterator_step$|"
                   r"last_overflow_iterator_step$)")
_PREFIXES = ("network.", "_backbone.", "acc_net.", "net.", "model.")


def convert_names(ref_params):
    """Reference (MindSpore cell) parameter names -> names of this package's ASRModel / ConformerEncoder state dict.
    Optimizer / loss-scale state is dropped; conv1d weights lose MindSpore's extra unit axis ((out, in, 1, k) -> (out, in, k))."""
    out = {}
    for name, arr in ref_params.items():
        while name.startswith(_PREFIXES):
            name = name[len(next(p for p in _PREFIXES if name.startswith(p))):]
        if _SKIP.match(name):
            continue
        n = name
        n = n.replace(".dense.", ".")                              # layers/dense.py:51
        n = re.sub(r"\.embed\.conv\.0\.conv2d\.", ".embed.conv1.", n)   # layers/subsampling.py:40-45, layers/conv2d.py:55
        n = re.sub(r"\.embed\.conv\.2\.conv2d\.", ".embed.conv2.", n)
        if ".conv1d." in n:                                          # layers/conv1d.py:73
            n = n.replace(".conv1d.", ".")
            if arr.ndim == 4 and arr.shape[2] == 1:
                arr = arr[:, :, 0, :]
        if ".conv_module.norm." in n:                                # layers/convolution.py:63 (BatchNorm1d)
            n = (n.replace(".norm.gamma", ".norm.weight").replace(".norm.beta", ".norm.bias")
                 .replace(".norm.moving_mean", ".norm.running_mean").replace(".norm.moving_variance", ".norm.running_var"))
        n = n.replace("decoder.embed.0.embedding_table", "decoder.embed.weight")  # models/conformer.py:555-558
        out[n] = arr
    return out


def to_reference_names(state, prefix=""):
    """Inverse of `convert_names` for the modules of this package: {state_dict name: tensor / ndarray} -> {the name and layout
    MindSpore gives the same parameter in the reference's cells: ndarray}, so that a file written from it loads with the reference's
    `load_param_into_net` (Dense.dense, Conv1d.conv1d with the unit axis, Conv2d.conv2d inside the SequentialCell, BatchNorm
    gamma / beta / moving_mean / moving_variance as Parameters, Embedding.embedding_table).  `num_batches_tracked` has no MindSpore
    counterpart and is dropped; GlobalCMVN's statistics are constructor data, never parameters (layers/cmvn.py)."""
    out = {}
    for k, v in state.items():
        if k.endswith("num_batches_tracked"):
            continue
        a = v.detach().float().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        n = k
        if ".embed.conv1." in n or ".embed.conv2." in n:
            n = n.replace(".embed.conv1.", ".embed.conv.0.conv2d.").replace(".embed.conv2.", ".embed.conv.2.conv2d.")
        elif ".conv_module." in n and ("pointwise_conv" in n or "depthwise_conv" in n):
            n = n.replace(".weight", ".conv1d.weight").replace(".bias", ".conv1d.bias")
            if a.ndim == 3:
                a = a[:, :, None, :]
        elif ".conv_module.norm." in n:
            n = (n.replace(".norm.weight", ".norm.gamma").replace(".norm.bias", ".norm.beta")
                 .replace("running_mean", "moving_mean").replace("running_var", "moving_variance"))
        elif n == "decoder.embed.weight":
            n = "decoder.embed.0.embedding_table"
        elif n.startswith("ctc.ctc_lo.") or n.endswith((".gamma", ".beta", "pos_bias_u", "pos_bias_v")):
            pass  # loss/ctc_loss.py builds a bare nn.Dense; LayerNorm and the attention biases are plain Parameters
        else:  # layers/dense.py: every other weight / bias sits inside the Dense wrapper
            n = n.replace(".weight", ".dense.weight").replace(".bias", ".dense.bias")
        out[prefix + n] = a
    return out


def read_epoch_num(path):
    """`int(param_dict.get("epoch_num", 0))` of examples/conformer/train.py:121 - the number of finished epochs the reference's
    ModelCheckpoint appends to every file (`append_info=[{"epoch_num": ...}]`, train.py:157-163)."""
    v = read_mindspore_ckpt(path).get("epoch_num")
    return int(np.asarray(v).reshape(-1)[0]) if v is not None else 0


def load_mindspore_checkpoint(module, path, strict=True):
    """load_checkpoint + load_param_into_net (examples/conformer/predict.py:66-67) for a module of this package.
    Returns (missing, unexpected) like torch's load_state_dict; raises on shape mismatches."""
    import torch

    params = convert_names(read_mindspore_ckpt(path))
    own = module.state_dict()
    state, unexpected = {}, []
    for k, v in params.items():
        if k not in own:
            unexpected.append(k)
            continue
        t = torch.from_numpy(np.ascontiguousarray(v))
        if tuple(t.shape) != tuple(own[k].shape):
            raise ValueError("shape mismatch for %s: checkpoint %s, module %s" % (k, tuple(t.shape), tuple(own[k].shape)))
        state[k] = t.to(own[k].dtype)
    missing = [k for k in own if k not in state and not k.endswith("num_batches_tracked")]
    if strict and (missing or unexpected):
        raise KeyError("checkpoint does not match the module: missing %s, unexpected %s" % (missing[:8], unexpected[:8]))
    # (ConformerEncoder / CTC drop their bf16 / packed weight copies in load_state_dict post-hooks)
    module.load_state_dict(state, strict=False)
    return missing, unexpected
