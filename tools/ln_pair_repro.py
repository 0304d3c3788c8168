#!/usr/bin/env python3
"""Round 4's two-queue victim as a PAIR: `layernorm_bwd_kernel` (the LayerNorm backward launch of the un-fused training step) on one
stream beside the launches round 4 had on the second hardware queue - the split-K grouped weight-gradient products
(`gemm_tn_group_kernel`) and the block's batched partial sums (`tn_reduce_batch_kernel`) - with the engine's real plan, arena and
shapes (cfg 4: 40 x 255 rows), every output compared bit for bit with the same launch run alone.

    python tools/ln_pair_repro.py [--iters 300] [--arm both|reduce|products|none] [--victims 3]

`tools/wg_hunt_loop.py --unfused-ln --arms wgsplit` reproduces the corruption at the level of the step (5 % of fresh processes, first
wrong tensor l10.norm_ff.g, profiles/r06_two_queue_hunt.txt); this tool asks which neighbour it takes."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--arm", default="both", choices=("both", "reduce", "products", "none", "q1"))
    ap.add_argument("--victims", type=int, default=3, help="victim launches per iteration on the main stream")
    ap.add_argument("--blocks", type=int, default=2)
    a = ap.parse_args()
    from mindaudio_amd import _host, _lib
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train import engine as E
    from mindaudio_amd.train import kernels as K

    dev = torch.device("cuda", 0)
    lib = _lib.load()
    torch.manual_seed(777)
    model = create_asr_model(80, 4233, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=a.blocks)).to(dev)
    E._TWO_QUEUE_REPRODUCER.update(wg_stream=True, split_k_sums_on_second_stream=True)
    eng = E.ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=4, dropout_rate=0.1, positional_dropout_rate=0.1, dw_group_blocks=0)
    eng._wg_from = 0
    eng.ffn_bwd_one_launch = eng.ln_bwd_fused = eng.ln_final_chained = False
    eng.block_tables = False
    eng._pack_plan = None
    eng._pack_weights()
    # one real step: builds the plan (arena, device item tables) at the cfg-4 shape
    rng = np.random.RandomState(1)
    b, t = 40, 1024
    xs = torch.from_numpy(rng.randn(b, t, 80).astype(np.float32)).to(dev)
    t2 = ((t - 3) // 2 + 1 - 3) // 2 + 1
    masks = torch.ones(b, 1, t2, device=dev)
    ys = torch.from_numpy(rng.randint(1, 4232, (b, 12)).astype(np.int32)).to(dev)
    yl = torch.full((b,), 12, dtype=torch.int32, device=dev)
    eng.step(xs, ys, None, None, None, None, masks, None, None, yl, None)
    torch.cuda.synchronize()
    plan = eng._dw_plan
    m, d = b * t2, 256
    items, block_item, n_blocks = plan["layers"][a.blocks - 1]  # the block whose sums run on the second queue
    arena = plan["arena"]
    arena.view(torch.float32)[:] = torch.randn(arena.numel() // 4, device=dev) * 0.01
    # the victim's operands: the LayerNorm backward of norm_ff of the block BELOW (the other half of the arena, as in the step)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(m, d, device=dev, generator=g) * 1.5 + 0.2
    gamma = 1 + 0.1 * torch.randn(d, device=dev, generator=g)
    dy = torch.randn(m, d, device=dev, generator=g).bfloat16()
    g0 = torch.randn(m, d, device=dev, generator=g)
    o, nbytes, parts = plan["off"]["norm_ff"]
    half = plan["half"]
    vic_parts = arena[((a.blocks - 2) & 1) * half + o:((a.blocks - 2) & 1) * half + o + nbytes].view(torch.float32)
    nxt = (1.0, 0.1, 1234, 77, None)
    # products for the second queue: eight split-K weight-gradient products of a block's shapes into the arena's other half
    prods = []
    if a.arm in ("both", "products"):
        for sfx in eng._DW_SUFFIXES:
            mo, no = eng.fp.w("l0." + sfx).shape
            oo, nb, _ = plan["off"][sfx]
            dyp = (torch.randn(m, mo, device=dev, generator=g) * 0.1).bfloat16()
            xp = (torch.randn(m, no, device=dev, generator=g) * 0.1).bfloat16()
            prods.append((dyp, xp, arena[((a.blocks - 1) & 1) * half + oo:((a.blocks - 1) & 1) * half + oo + nb]))
    main, side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    side_ptr = __import__("ctypes").c_void_p(side.cuda_stream)
    grad_lo, grad_hi = eng.fp.span(eng.layer_names[a.blocks - 1])

    def victim():
        gg = g0.clone()
        outs = []
        for k in range(a.victims):
            _, dn = K.layernorm_bwd_next(x, gamma, dy, gg, None, None, nxt, partials=vic_parts)
            outs.append(dn)
        return gg, outs

    def aggressor():
        if prods:
            K.gemm_tn_partial_group(prods, with_colsum=True)
        if a.arm in ("both", "reduce", "q1"):
            _lib.check(lib.ma_reduce_splits_batch_f32(items.data_ptr(), block_item.data_ptr(), n_blocks, _host.current_stream_ptr()),
                       "reduce")

    # references: each side alone
    with torch.cuda.stream(main):
        g_ref, dn_ref = victim()
        p_ref = vic_parts.clone()
    torch.cuda.synchronize()
    bad_v = bad_a = 0
    first = None
    eng.fp.grad.zero_()
    with torch.cuda.stream(side):
        prev = _host.swap_pinned(side_ptr)
        try:
            aggressor()
        finally:
            _host.swap_pinned(prev)
    torch.cuda.synchronize()
    grad_ref = eng.fp.grad[grad_lo:grad_hi].clone()
    t0 = time.time()
    for it in range(a.iters):
        eng.fp.grad.zero_()
        vic_parts.zero_()
        torch.cuda.synchronize()
        if a.arm != "none":
            with torch.cuda.stream(side):
                prev = _host.swap_pinned(side_ptr)
                try:
                    aggressor()
                finally:
                    _host.swap_pinned(prev)
        with torch.cuda.stream(main):
            gg, dns = victim()
        torch.cuda.synchronize()
        ok_v = torch.equal(gg, g_ref) and all(torch.equal(p_, q_) for p_, q_ in zip(dns, dn_ref)) and torch.equal(vic_parts, p_ref)
        ok_a = a.arm == "none" or torch.equal(eng.fp.grad[grad_lo:grad_hi], grad_ref)
        if not ok_v:
            bad_v += 1
            if first is None:
                dg = (gg != g_ref)
                rows = dg.any(1).nonzero().flatten()
                pv = (vic_parts != p_ref).nonzero().flatten()
                first = dict(iter=it, g_rows_wrong=int(rows.numel()), g_first_rows=rows[:8].tolist(),
                             g_cols_of_first_row=dg[rows[0]].nonzero().flatten()[:16].tolist() if rows.numel() else [],
                             parts_wrong=int(pv.numel()), parts_first=pv[:16].tolist(),
                             dn_wrong=[int((p_ != q_).sum()) for p_, q_ in zip(dns, dn_ref)])
        bad_a += not ok_a
    print(json.dumps(dict(arm=a.arm, iters=a.iters, victim_bad=bad_v, aggressor_bad=bad_a, first=first,
                          seconds=round(time.time() - t0, 1), hwq=os.environ.get("GPU_MAX_HW_QUEUES"))))


if __name__ == "__main__":
    main()
