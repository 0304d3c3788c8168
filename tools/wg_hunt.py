"""Development: ONE fresh-process sample of the cfg-4 training step for the two-hardware-queue corruption hunt.

    python tools/wg_hunt.py --mode {single,wg,serial,foreign} [--steps 3] [--same-batch] --out sums.json

Every step gets a DIFFERENT batch and (through the engine's per-step seed) different dropout masks, and the output is a checksum of
every gradient tensor after every step - so a read of stale data (last step's, or another engine's, near-identical values) shows as a
mismatch against the single-stream twin instead of hiding behind identical bytes; tools/wg_hunt_loop.py compares the sums of each
sample with the `single` reference and names the first tensor of the backward chain that differs.

modes: single  = one stream (the reference);
       wg      = the engine's second stream from the FIRST step on (weight-gradient products + batched sums beside the main chain);
       serial  = the same two streams, but the main stream waits for every block's products at once: two hardware queues, no
                 concurrent execution (ordering / visibility between queues vs concurrency);
       rccl    = one stream for the library + RCCL's stream: every gradient bucket through a world-1 NCCL group as ReduceOp.AVG (a real
                 RCCL kernel per bucket, the identity): the concurrency N > 1 training creates;
       wgsplit = `wg` with round 3's products (one split-K grid + batched sum per block) instead of the direct 256 x 256-tile groups;
       foreign = the engine on ONE stream (every product of the library on the main stream), while a second stream runs torch's own
                 matmuls and copies at the points where `wg` would run the products (the library's kernels vs any concurrent kernel).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

V, T, BLOCKS = 4233, 1024, 12


def batch(b, seed):
    rng = np.random.RandomState(seed)
    xs = torch.from_numpy(rng.randn(b, T, 80).astype(np.float32))
    lens = rng.randint(int(0.7 * T), T + 1, b)
    lens[0] = T
    t2 = ((T - 3) // 2 + 1 - 3) // 2 + 1
    sub = torch.zeros(b, 1, t2)
    for i, n in enumerate(lens):
        sub[i, 0, :(n - 1) // 4] = 1
        xs[i, n:] = 0
    ylens = torch.from_numpy(rng.randint(5, 31, b).astype(np.int32))
    ys = torch.full((b, 30), -1, dtype=torch.int32)
    for i, n in enumerate(ylens.tolist()):
        ys[i, :n] = torch.from_numpy(rng.randint(1, V - 1, n).astype(np.int32))
    return xs, ys, sub, ylens


def tensor_sums(eng):
    """int64 wrap-around sum of the raw 32-bit patterns of every gradient tensor (one cumsum + one gather)."""
    g = eng.fp.grad
    cs = torch.cumsum(g.view(torch.int32).to(torch.int64), 0)
    names, lo, hi = [], [], []
    for name, (off, shape, n) in eng.fp.index.items():
        names.append(name)
        lo.append(off)
        hi.append(off + n - 1)
    lo_t = torch.tensor(lo, device=g.device)
    hi_t = torch.tensor(hi, device=g.device)
    first = torch.where(lo_t > 0, cs[(lo_t - 1).clamp(min=0)], torch.zeros_like(lo_t))
    return names, (cs[hi_t] - first).tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="wg", choices=("single", "wg", "serial", "foreign", "wgsplit", "rccl"))
    ap.add_argument("--diag", action="store_true", help="on a mismatch with an in-process single-stream twin: which elements differ")
    ap.add_argument("--tn-lds", type=int, default=0, help="ma_debug_tn_group_lds: dynamic LDS of the split-K grouped kernel (81920 = "
                    "its two workgroups per CU take the whole LDS: no co-residency with another kernel's workgroups)")
    ap.add_argument("--reduce-on-main", action="store_true",
                    help="wgsplit only: the batched sums on the MAIN stream (one block late), only the grouped products on the second")
    ap.add_argument("--group", type=int, default=6, help="dw_group_blocks of the engine (0 = split-K products per block)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--unfused-ln", action="store_true",
                    help="round 4's launch set: LayerNorm backwards as launches of their own (ffn_bwd_one_launch, ln_bwd_fused and "
                         "ln_final_chained off) - layernorm_bwd_kernel is back at the place in the chain where round 4's victim ran")
    ap.add_argument("--same-batch", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train import engine as _engine
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    dev = torch.device("cuda", 0)
    if a.mode == "rccl":
        # one stream for the library, RCCL's own stream beside it: the 14 gradient buckets of every step go through ProcessGroupNCCL at
        # world size 1 as ReduceOp.AVG (a real RCCL device kernel per bucket, the identity) - what N > 1 training does to the step
        import socket

        import torch.distributed as dist

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(777)
    model = create_asr_model(80, V, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=BLOCKS)).to(dev)
    # wg / serial: the engine's second stream with its default products (direct 256 x 256-tile groups); wgsplit: round 3's form - one
    # split-K grid + batched sum per block on the second stream
    # (the two-queue path is not a constructor option any more: the reproducer hook of train/engine.py)
    _engine._TWO_QUEUE_REPRODUCER.update(wg_stream=a.mode in ("wg", "serial", "wgsplit"), split_k_sums_on_second_stream=a.mode == "wgsplit")
    eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=4, dropout_rate=0.1, positional_dropout_rate=0.1,
                                dw_group_blocks=0 if a.mode == "wgsplit" else a.group, force_collective=a.mode == "rccl")
    eng._wg_from = 0
    if a.unfused_ln:
        eng.ffn_bwd_one_launch = eng.ln_bwd_fused = eng.ln_final_chained = False
        eng.block_tables = False
        eng._pack_plan = None  # (the backward's packed weight forms depend on the switches)
        eng._pack_weights()
    if a.tn_lds:
        from mindaudio_amd import _lib as L

        L.check(L.load().ma_debug_tn_group_lds(a.tn_lds), "tn_group_lds")
    orig_done = eng._layer_done
    if a.mode == "wgsplit" and a.reduce_on_main:
        from mindaudio_amd import _host as H
        from mindaudio_amd import _lib as L

        pend = []

        def reduce_main(li, ev):
            items, block_item, n_blocks = eng._dw_cur["layers"][li]
            eng._main.wait_event(ev)
            L.check(L.load().ma_reduce_splits_batch_f32(items.data_ptr(), block_item.data_ptr(), n_blocks, eng._main.cuda_stream), "reduce")
            eng.reducer.launch(*eng.fp.span(eng.layer_names[li]))

        def done_tnonly(li):
            if eng._wg is None:
                return orig_done(li)
            eng._wg.wait_event(eng._wg_event().record_on(eng._main))
            prev = H.swap_pinned(eng._wg_ptr)
            try:
                eng.K.gemm_tn_partial_group(eng._wg_queue, with_colsum=True)
            finally:
                H.swap_pinned(prev)
            eng._wg_keep.extend(eng._wg_queue)
            eng._wg_queue.clear()
            e = eng._wg_event()
            e.ev.record(eng._wg)
            if pend:  # the previous block's sums: its products have had a whole block of main-stream work to finish beside
                reduce_main(*pend.pop())
            pend.append((li, e.ev))
            if li == 0:
                reduce_main(*pend.pop())
        eng._layer_done = done_tnonly
    if a.mode == "serial":
        def done_serial(li):
            orig_done(li)
            if eng._wg is not None and li in eng._wg_done:
                eng._main.wait_event(eng._wg_done[li].ev)
        eng._layer_done = done_serial
    elif a.mode == "foreign":
        side = torch.cuda.Stream(device=dev)
        ja = torch.randn(10200, 2048, device=dev).to(torch.bfloat16)
        jb = torch.randn(10200, 256, device=dev).to(torch.bfloat16)
        jc = torch.empty(2048, 256, device=dev, dtype=torch.bfloat16)
        j1 = torch.randn(20 << 20, device=dev)
        j2 = torch.empty_like(j1)
        evs = [torch.cuda.Event() for _ in range(BLOCKS)]

        def done_foreign(li):
            main = torch.cuda.current_stream()
            evs[li].record(main)
            side.wait_event(evs[li])
            with torch.cuda.stream(side):
                for _ in range(8):
                    torch.mm(ja.t(), jb, out=jc)
                j2.copy_(j1)
            orig_done(li)
        eng._layer_done = done_foreign
    out = dict(mode=a.mode, steps=[], env={k: os.environ[k] for k in ("GPU_MAX_HW_QUEUES",) if k in os.environ})
    grad0 = None
    if a.mode == "rccl":
        # the library on a stream of its own: RCCL's stream can share the hardware queue of the legacy default stream (seen in a kernel
        # trace), in which case nothing would run beside the backward pass
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(side)
    for s in range(a.steps):
        xs, ys, sub, yl = batch(40, 43 if a.same_batch else 43 + s)
        cols = (xs.to(dev), ys.to(dev), None, None, None, None, sub.to(dev), None, None, yl.to(dev), None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, cond, scale, overflow, lr = eng.step(*cols)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        names, sums = tensor_sums(eng)
        if a.diag and s == 0:
            grad0 = eng.fp.grad.clone()
        out["steps"].append(dict(loss=float(loss), overflow=bool(overflow), sums=sums, host_ms=round(t_host * 1e3, 2),
                                 ms=round(t_all * 1e3, 2)))
        out["names"] = names
    if a.mode == "foreign":
        torch.cuda.synchronize()
    if a.diag:
        # the single-stream twin of step 0 in THIS process (a later engine of a process has never failed), element by element
        torch.manual_seed(777)
        model2 = create_asr_model(80, V, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=BLOCKS)).to(dev)
        eng2 = ConformerCTCTrainStep(model2, base_lr=1e-3, warmup_steps=4, dropout_rate=0.1, positional_dropout_rate=0.1,
                                     dw_group_blocks=0 if a.mode == "wgsplit" else a.group)
        xs, ys, sub, yl = batch(40, 43)
        eng2.step(xs.to(dev), ys.to(dev), None, None, None, None, sub.to(dev), None, None, yl.to(dev), None)
        g2 = eng2.fp.grad
        diag = []
        for name, (off, shape, n) in eng.fp.index.items():
            x, y = grad0[off:off + n], g2[off:off + n]
            ne = (x != y)
            k = int(ne.sum())
            if k:
                idx = ne.nonzero().flatten()
                cols = shape[-1] if len(shape) > 1 else n
                d = (x - y).abs()
                diag.append(dict(name=name, shape=list(shape), n_diff=k, max_abs=float(d.max()), ref_max=float(y.abs().max()),
                                 first=idx[:6].tolist(), last=int(idx[-1]), rows_hit=int(torch.unique(idx // cols).numel()),
                                 cols_hit=int(torch.unique(idx % cols).numel()), nonfinite=int((~torch.isfinite(x)).sum())))
        out["diag"] = diag
    if a.mode == "rccl":
        dist.destroy_process_group()
    txt = json.dumps(out)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)
    else:
        print(txt)


if __name__ == "__main__":
    main()
