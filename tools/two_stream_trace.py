"""Development: two encoder forwards concurrently on two streams, every launch's outputs cloned into a trace; the first entry of a
trace that differs from the same forward run alone names the launch that is not independent of what runs beside it.
    python tools/two_stream_trace.py [fused_front=0|1] [blocks]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mindaudio_amd import ops
from mindaudio_amd.models import ConformerEncoder

fused_front = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.manual_seed(3)
enc = ConformerEncoder(80, 256, 4, 2048, blocks).eval().cuda()
enc.subsample_fused = fused_front
enc.prepare()
TRACE = None
NAMES = ["subsample_fused", "subsample_conv1", "conv2d_3x3s2_packed", "gemm_rows_packed", "ffn_packed_qkv", "relpos_attention",
         "attn_out_convmodule", "ffn_packed_pair", "ffn_packed", "gemm", "gemm_packed", "layernorm"]
for name in NAMES:
    if not hasattr(ops, name):
        continue

    def wrap(fn, name=name):
        def call(*a, **k):
            r = fn(*a, **k)
            if TRACE is not None:
                outs = [r] if torch.is_tensor(r) else [t for t in (r or ()) if torch.is_tensor(t)]
                ins = [t for t in list(a) + list(k.values()) if torch.is_tensor(t) and t.dtype == torch.float32 and t.dim() == 2 and t.shape[0] > 4096]
                TRACE.append((name, [o.clone() for o in outs] + [i.clone() for i in ins]))
            return r
        return call
    setattr(ops, name, wrap(getattr(ops, name)))

b, frames = 32, 1000
t2 = ((frames - 3) // 2 + 1 - 3) // 2 + 1
xs = [torch.randn(b, frames, 80, device="cuda") for _ in range(2)]
m = torch.ones(b, 1, t2, device="cuda")
enc(xs[0], m)  # (the first forward of a shape also computes the cached positional projections)
want = []
for x in xs:
    TRACE = []
    enc(x, m)
    want.append(TRACE)
TRACE = None
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
cur = torch.cuda.current_stream()
import threading

first = {}
for it in range(12):
    traces = [[], []]

    def work(i):
        global TRACE
        with torch.cuda.stream(streams[i]):
            enc(xs[i], m)

    # (one host thread issues both chains alternately would serialise the enqueue; the trace list is chosen per call instead)
    for i in range(2):
        streams[i].wait_stream(cur)
    # interleave: run forward i with TRACE bound to its list
    for i in range(2):
        TRACE = traces[i]
        with torch.cuda.stream(streams[i]):
            enc(xs[i], m)
    TRACE = None
    for s_ in streams:
        cur.wait_stream(s_)
    torch.cuda.synchronize()
    for i in range(2):
        for k, ((n1, t1), (n2, t2_)) in enumerate(zip(traces[i], want[i])):
            assert n1 == n2 and len(t1) == len(t2_), (k, n1, n2)
            if not all(torch.equal(a, b_) for a, b_ in zip(t1, t2_)):
                which = [j for j, (a, b_) in enumerate(zip(t1, t2_)) if not torch.equal(a, b_)]
                d = max(float((a.float() - b_.float()).abs().max()) for a, b_ in zip(t1, t2_))
                key = (k, n1)
                first[key] = first.get(key, 0) + 1
                print("iteration %d stream %d: first difference at launch %d (%s), tensors %s, max |diff| %.4g" % (it, i, k, n1, which, d), flush=True)
                a_, b_ = t1[which[0]], t2_[which[0]]
                ne = (a_ != b_)
                idx = ne.nonzero()
                print("    shape %s, %d elements differ; index ranges per dim: %s; first %s last %s; got there %s" % (
                    tuple(a_.shape), int(ne.sum()), [(int(idx[:, d_].min()), int(idx[:, d_].max())) for d_ in range(idx.shape[1])],
                    idx[0].tolist(), idx[-1].tolist(), a_[tuple(idx[0].tolist())].item()), flush=True)
                for j_ in range(0, min(len(idx), 4000), max(1, min(len(idx), 4000) // 8)):
                    ii = idx[j_].tolist()
                    ch0 = ii[-1] // 8 * 8
                    print("      at %s: got %s | want %s" % (ii, [round(float(v), 3) for v in a_[tuple(ii[:-1])][ch0:ch0 + 8]],
                                                            [round(float(v), 3) for v in b_[tuple(ii[:-1])][ch0:ch0 + 8]]), flush=True)
                chs = idx[:, -1]
                print("      channel histogram mod 8: %s ; f1 histogram mod 8: %s" % (torch.bincount(chs % 8, minlength=8).tolist(),
                      torch.bincount(idx[:, 2] % 8, minlength=8).tolist()), flush=True)
                break
print("first differing launches:", first)
