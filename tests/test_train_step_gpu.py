"""GPU parity of the hand-written training step (mindaudio_amd.train.engine) against PyTorch-CPU autograd of the
oracle model (oracle/conformer_oracle.py, float32): loss, every parameter gradient, the Adam update with the
reference's loss-scale semantics, and BatchNorm running statistics.  The device multiplies in bf16 (float32
accumulation), the oracle in float32: gradients are compared per tensor by relative RMS error."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def build(vocab=97, blocks=2, seed=5, cmvn=True):
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from oracle import conformer_oracle as C

    torch.manual_seed(seed)
    mean = istd = None
    kw = {}
    if cmvn:
        mean, istd = torch.randn(80) * 0.5, torch.rand(80) + 0.5
        kw = dict(cmvn_mean=mean, cmvn_istd=istd)
    ref_enc = C.ConformerEncoder(80, 256, 4, 2048, blocks, dropout_rate=0.0, positional_dropout_rate=0.0, **kw)
    ref_ctc = C.CTC(vocab, 256)
    with torch.no_grad():
        for m in ref_enc.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.8, 1.2)
                m.bias.normal_(0, 0.1)
            if isinstance(m, C.LayerNorm):
                m.gamma.uniform_(0.8, 1.2)
                m.beta.normal_(0, 0.1)
    model = create_asr_model(80, vocab, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=blocks),
                             global_cmvn=(mean, istd) if cmvn else None)
    missing, unexpected = model.encoder.load_state_dict(ref_enc.state_dict(), strict=False)
    assert not missing and not [k for k in unexpected if "cmvn" not in k]
    model.ctc.load_state_dict(ref_ctc.state_dict())
    return ref_enc.train(), ref_ctc.train(), model.cuda()


def batch(b=3, tlen=131, vocab=97, seed=9, shorter=(0, 30, 61), ylens=(9, 6, 4)):
    from oracle import conformer_oracle as C

    g = torch.Generator().manual_seed(seed)
    xs = torch.randn(b, tlen, 80, generator=g)
    lens = [tlen - d for d in shorter][:b]
    mask = torch.zeros(b, 1, tlen)
    for i, n in enumerate(lens):
        mask[i, 0, :n] = 1
        xs[i, n:] = 0
    sub = C.subsample_mask(mask)
    ys_lens = torch.tensor(list(ylens)[:b], dtype=torch.int32)
    ys = torch.full((b, 9), -1, dtype=torch.int32)
    for i, n in enumerate(ys_lens.tolist()):
        ys[i, :n] = torch.randint(1, vocab, (n,), generator=g, dtype=torch.int32)
    return xs, ys, sub, ys_lens


def oracle_loss(ref_enc, ref_ctc, xs, ys, sub, ys_lens):
    out, m = ref_enc(xs, sub)
    hlens = m.reshape(m.shape[0], -1).sum(1).to(torch.int32)
    return ref_ctc(out, hlens, ys.clamp(min=0).long(), ys_lens.long())


def rel_rms(got, want):
    return float((got.float().cpu() - want).norm() / (want.norm() + 1e-12))


def test_forward_backward_matches_oracle_autograd():
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    ref_enc, ref_ctc, model = build()
    xs, ys, sub, ys_lens = batch()
    loss_ref = oracle_loss(ref_enc, ref_ctc, xs, ys, sub, ys_lens)
    loss_ref.backward()
    eng = ConformerCTCTrainStep(model, dropout_rate=0.0, positional_dropout_rate=0.0)
    loss = eng.forward_backward(xs.cuda(), ys.cuda(), sub.cuda(), ys_lens.cuda(), grad_scale=1.0)
    assert abs(float(loss) - float(loss_ref.detach())) <= 2e-2 * abs(float(loss_ref.detach()))
    grads = eng.gradients()
    want = {"encoder." + n: p.grad for n, p in ref_enc.named_parameters()}
    want.update({"ctc." + n: p.grad for n, p in ref_ctc.named_parameters()})
    assert set(grads) == set(want)
    worst = {}
    for name, gw in want.items():
        # identically-zero gradients (rounding noise on both sides): the depthwise bias feeds a BatchNorm; a key
        # bias shifts every score of a row by the same amount, which softmax ignores
        if "depthwise_conv.bias" in name or "linear_k.bias" in name:
            assert float(grads[name].abs().max()) < 1e-3 * float(max(p.abs().max() for p in want.values()))
            continue
        worst[name] = rel_rms(grads[name], gw)
    bad = {k: round(v, 4) for k, v in worst.items() if v > 6e-2}
    assert not bad, bad
    assert sum(worst.values()) / len(worst) < 2.5e-2
    # BatchNorm running statistics moved as nn.BatchNorm1d's do
    for l_ref, m_, v_ in zip(ref_enc.encoders, eng.bn_mean, eng.bn_var):
        assert rel_rms(m_, l_ref.conv_module.norm.running_mean) < 2e-2
        assert rel_rms(v_, l_ref.conv_module.norm.running_var) < 2e-2


def test_float32_mode_gradients_match_oracle_autograd_tightly():
    """compute_type=float32 (the reference's default, models/conformer.py:61): same tape and backward as the bf16 mode, float32
    activations and products -> every parameter gradient within 2e-4 of PyTorch-CPU float32 autograd of the oracle (the bf16
    mode's bound above is 6e-2)."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    ref_enc, ref_ctc, model = build()
    xs, ys, sub, ys_lens = batch()
    loss_ref = oracle_loss(ref_enc, ref_ctc, xs, ys, sub, ys_lens)
    loss_ref.backward()
    eng = ConformerCTCTrainStep(model, dropout_rate=0.0, positional_dropout_rate=0.0, compute_type=torch.float32)
    loss = eng.forward_backward(xs.cuda(), ys.cuda(), sub.cuda(), ys_lens.cuda(), grad_scale=1.0)
    assert abs(float(loss) - float(loss_ref.detach())) <= 2e-6 * abs(float(loss_ref.detach()))
    grads = eng.gradients()
    want = {"encoder." + n: p.grad for n, p in ref_enc.named_parameters()}
    want.update({"ctc." + n: p.grad for n, p in ref_ctc.named_parameters()})
    assert set(grads) == set(want)
    gmax = float(max(p.abs().max() for p in want.values()))
    worst = {}
    for name, gw in want.items():
        if "depthwise_conv.bias" in name or "linear_k.bias" in name:  # identically zero: float32 noise on both sides
            assert float(grads[name].abs().max()) < 1e-5 * gmax
            continue
        worst[name] = rel_rms(grads[name], gw)
    bad = {k: v for k, v in worst.items() if v > 2e-4}
    print("float32 mode: worst per-tensor relative gradient error %.2e" % max(worst.values()))
    assert not bad, bad
    for l_ref, m_, v_ in zip(ref_enc.encoders, eng.bn_mean, eng.bn_var):
        assert rel_rms(m_, l_ref.conv_module.norm.running_mean) < 1e-5
        assert rel_rms(v_, l_ref.conv_module.norm.running_var) < 1e-5


@pytest.mark.parametrize("mode", ["bf16", "float32"])
def test_chunk_masks_in_training_match_oracle_autograd(mode):
    """The streaming configuration trains with (B, T', T') attention masks (utils/mask.py:201-271 add_optional_chunk_mask;
    models/conformer.py:251-252 hands them to every block): loss and every parameter gradient against autograd of the oracle
    given the same masks, in both compute types."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    ref_enc, ref_ctc, model = build()
    xs, ys, sub, ys_lens = batch()
    t2 = sub.shape[-1]
    idx = torch.arange(t2)
    chunk = ((idx[None, :] // 8) <= (idx[:, None] // 8)) & ((idx[None, :] // 8) >= (idx[:, None] // 8) - 2)  # chunk 8, 2 left chunks
    chunk_masks = (chunk[None] & (sub > 0)).float()  # (B, T', T'): masks & chunk_masks (mask.py:262-267)
    out, m = ref_enc(xs, sub, chunk_masks)
    hlens = m.reshape(m.shape[0], -1).sum(1).to(torch.int32)
    loss_ref = ref_ctc(out, hlens, ys.clamp(min=0).long(), ys_lens.long())
    loss_full = float(oracle_loss(ref_enc, ref_ctc, xs, ys, sub, ys_lens).detach())
    assert abs(loss_full - float(loss_ref.detach())) > 1e-4 * abs(loss_full)  # (the chunk masks do change the loss: 6e-4 here)
    loss_ref.backward()
    kw = dict(compute_type=torch.float32) if mode == "float32" else {}
    eng = ConformerCTCTrainStep(model, dropout_rate=0.0, positional_dropout_rate=0.0, **kw)
    loss = eng.forward_backward(xs.cuda(), ys.cuda(), sub.cuda(), ys_lens.cuda(), xs_chunk_masks=chunk_masks.cuda(),
                                grad_scale=1.0)
    ltol, gtol = (2e-6, 2e-4) if mode == "float32" else (2e-2, 6e-2)
    assert abs(float(loss) - float(loss_ref.detach())) <= ltol * abs(float(loss_ref.detach()))
    grads = eng.gradients()
    want = {"encoder." + n: p.grad for n, p in ref_enc.named_parameters()}
    want.update({"ctc." + n: p.grad for n, p in ref_ctc.named_parameters()})
    gmax = float(max(p.abs().max() for p in want.values()))
    worst = {}
    for name, gw in want.items():
        if "depthwise_conv.bias" in name or "linear_k.bias" in name:  # identically zero (see above)
            assert float(grads[name].abs().max()) < 1e-3 * gmax
            continue
        worst[name] = rel_rms(grads[name], gw)
    bad = {k: v for k, v in worst.items() if v > gtol}
    assert not bad, bad


@pytest.mark.parametrize("mode", ["bf16", "float32"])
@pytest.mark.parametrize("n_steps", [6, 50])
def test_train_steps_follow_the_oracle_loss_curve(n_steps, mode):
    """Optimizer steps on one batch: Adam + ASRWarmupLR + dynamic loss scale vs the same recipe in PyTorch (float32 oracle).
    SURVEY 8d / north_star: >= 50 steps within 1e-4.  compute_type=float32 (the reference's default) meets it; the bf16
    throughput mode (the reference's own mixed-precision compute_type is float16) follows the curve to bf16 round-off, and its
    measured deviation is asserted and reported."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep, asr_warmup_lr

    ref_enc, ref_ctc, model = build(seed=6, cmvn=False)
    xs, ys, sub, ys_lens = batch(seed=10)
    params = list(ref_enc.parameters()) + list(ref_ctc.parameters())
    opt = torch.optim.Adam(params, lr=1.0, betas=(0.9, 0.999), eps=1e-8)
    # 6 steps at an aggressive rate (the loss falls 4x: every part of the update rule matters); 50 steps at a gentle one, where
    # two trajectories that differ by bf16 round-off stay comparable instead of diverging chaotically
    warm, base = (8, 2e-3) if n_steps == 6 else (25, 2e-4)
    eng = ConformerCTCTrainStep(model, base_lr=base, warmup_steps=warm, dropout_rate=0.0, positional_dropout_rate=0.0,
                                compute_type=torch.float32 if mode == "float32" else None)
    cols = (xs.cuda(), ys.cuda(), None, None, None, None, sub.cuda(), None, None, ys_lens.cuda(), None)
    got, want = [], []
    for step in range(n_steps):
        lr = asr_warmup_lr(step, base, warm)
        for gq in opt.param_groups:
            gq["lr"] = lr
        opt.zero_grad()
        l_ref = oracle_loss(ref_enc, ref_ctc, xs, ys, sub, ys_lens)
        l_ref.backward()
        opt.step()
        want.append(float(l_ref))
        loss, cond, scale, overflow, lr_dev = eng.step(*cols)
        assert not overflow and not cond and scale == 1024.0 and abs(lr_dev - lr) < 1e-12
        got.append(float(loss))
    assert want[-1] < want[0]  # the recipe learns on this batch
    dev = max(abs(a - b_) / abs(b_) for a, b_ in zip(got, want))
    print("%s loss curve over %d steps: %.3f -> %.3f, max relative deviation from the float32 oracle %.2e"
          % (mode, n_steps, want[0], want[-1], dev))
    if mode == "float32":
        tol = 1e-4                           # the north-star tolerance
    else:
        tol = 3e-2 if n_steps == 6 else 5e-3  # bf16 matmuls, measured: 8e-3 / 1.3e-3
    for a, b_ in zip(got, want):
        assert abs(a - b_) <= tol * abs(b_), (got, want)
    # lr is 0 at step 0 (scheduler_factory.py:44-50): the first step must not move the weights
    assert abs(got[0] - got[1]) <= 2e-2 * abs(got[0]) or got[1] < got[0]


def test_overflow_skips_update_and_halves_scale():
    from mindaudio_amd.train.engine import ConformerCTCTrainStep, asr_warmup_lr

    _, _, model = build(blocks=1, seed=7, cmvn=False)
    xs, ys, sub, ys_lens = batch(b=2, seed=11)
    eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=2, dropout_rate=0.1, positional_dropout_rate=0.1,
                                scale_window=2)
    cols = (xs.cuda(), ys.cuda(), None, None, None, None, sub.cuda(), None, None, ys_lens.cuda(), None)
    eng.step(*cols)
    before = eng.fp.master.clone()
    xs_bad = xs.clone()
    xs_bad[0, 3, 5] = float("inf")
    loss, cond, scale, overflow, lr_bad = eng.step(xs_bad.cuda(), *cols[1:])
    assert overflow and cond and scale == 1024.0 and eng.scaler.scale == 512.0
    assert torch.equal(eng.fp.master, before)
    # counters (train_one_step.py:44-46): get_lr() ran on the skipped step too -> the LR index moved on; the optimizer did
    # not run -> Adam's bias-correction count did not
    assert (eng.global_step, eng.applied_steps) == (2, 1) and lr_bad == asr_warmup_lr(1, 1e-3, 2)
    _, _, scale, overflow, lr3 = eng.step(*cols)
    assert not overflow and scale == 512.0 and lr3 == asr_warmup_lr(2, 1e-3, 2)
    assert (eng.global_step, eng.applied_steps) == (3, 2)
    eng.step(*cols)
    assert eng.scaler.scale == 1024.0  # two clean steps (scale_window=2) double it again
    assert not torch.equal(eng.fp.master, before)
    eng.sync_to_module()
    assert torch.equal(model.encoder.after_norm.gamma.detach(), eng.fp.p("after_norm.g"))


@pytest.mark.parametrize("len_norm", [False, True])
def test_hybrid_ctc_attention_loss_and_gradients_match_oracle(len_norm):
    """ctc_weight 0.3 + TransformerDecoder + label smoothing 0.1 (conformer.yaml defaults, asr_model.py:75-186); len_norm =
    length_normalized_loss (asr_model.py:61: the attention loss divided by the token count instead of the batch size)."""
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train.engine import ConformerCTCTrainStep
    from oracle import conformer_oracle as C

    vocab, blocks, dblocks = 97, 1, 2
    torch.manual_seed(31)
    ref_enc = C.ConformerEncoder(80, 256, 4, 2048, blocks, dropout_rate=0.0, positional_dropout_rate=0.0).train()
    ref_ctc = C.CTC(vocab, 256).train()
    ref_dec = C.TransformerDecoder(vocab, 256, 4, 512, dblocks, 0.0, 0.0).train()
    with torch.no_grad():
        for mod in list(ref_enc.modules()) + list(ref_dec.modules()):
            if isinstance(mod, C.LayerNorm):
                mod.gamma.uniform_(0.8, 1.2)
                mod.beta.normal_(0, 0.1)
    model = create_asr_model(80, vocab, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=blocks),
                             ctc_weight=0.3, decoder_conf=dict(attention_heads=4, linear_units=512, num_blocks=dblocks),
                             lsm_weight=0.1, length_normalized_loss=len_norm)
    missing, unexpected = model.encoder.load_state_dict(ref_enc.state_dict(), strict=False)
    assert not missing and not [k for k in unexpected if "cmvn" not in k]
    model.ctc.load_state_dict(ref_ctc.state_dict())
    missing, unexpected = model.decoder.load_state_dict(ref_dec.state_dict(), strict=False)
    assert not missing and not unexpected
    model = model.cuda()
    # (rounds 5 and 6 shipped this test cut in two by a stray `return` at this point - everything below sat in a shadowed helper and
    # never ran; found and repaired at the end of round 6)
    cols = _hybrid_cols(vocab - 1, batch(vocab=vocab - 1, seed=12))
    loss_ref, acc_ref, lc_ref, la_ref = C.hybrid_loss(ref_enc, ref_ctc, ref_dec, cols, 0.3, 0.1, len_norm)
    loss_ref.backward()
    eng = ConformerCTCTrainStep(model, dropout_rate=0.0, positional_dropout_rate=0.0)
    model.decoder.dropout_rate = model.decoder.positional_dropout_rate = 0.0
    dev = [c.cuda() if c is not None else None for c in cols]
    loss = eng.forward_backward(dev[0], dev[1], dev[6], dev[9], None, 1.0, ys_in_pad=dev[2], ys_out_pad=dev[3],
                                ys_sub_masks=dev[7], ys_masks=dev[8])
    assert abs(float(eng.last_loss_ctc) - float(lc_ref.detach())) <= 2e-2 * abs(float(lc_ref.detach()))
    assert abs(float(eng.last_loss_att) - float(la_ref.detach())) <= 2e-2 * abs(float(la_ref.detach()))
    assert abs(float(loss) - float(loss_ref.detach())) <= 2e-2 * abs(float(loss_ref.detach()))
    assert abs(float(eng.last_acc) - float(acc_ref)) <= 0.05
    grads = eng.gradients()
    want = {"encoder." + n: p.grad for n, p in ref_enc.named_parameters()}
    want.update({"ctc." + n: p.grad for n, p in ref_ctc.named_parameters()})
    want.update({"decoder." + n: p.grad for n, p in ref_dec.named_parameters()})
    assert set(grads) == set(want)
    gmax = float(max(p.abs().max() for p in want.values()))
    worst = {}
    for name, gw in want.items():
        # identically-zero gradients: biases in front of a BatchNorm / of attention keys
        if "depthwise_conv.bias" in name or "linear_k.bias" in name:
            assert float(grads[name].abs().max()) < 1e-3 * gmax
            continue
        worst[name] = rel_rms(grads[name], gw)
    bad = {k: round(v, 4) for k, v in worst.items() if v > 6e-2}
    assert not bad, bad
    assert sum(worst.values()) / len(worst) < 2.5e-2
    # one optimizer step runs end to end with the 11 collate columns
    out = eng.step(*dev)
    assert len(out) == 5 and not out[1] and not out[3] and float(out[0]) > 0


@pytest.mark.parametrize("mode", ["bf16", "float32"])
def test_hybrid_gradients_with_labels_longer_than_one_query_tile(mode):
    """Transcripts of more than 31 tokens (AISHELL's longest; token_max_length is 200 in conformer.yaml): the decoder's attention
    kernels take 32 queries per launch and refused longer labels until round 6 - the hybrid step then died on the first such batch.
    Two query tiles here (L + 1 = 41): loss and every gradient against the oracle's autograd."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep
    from oracle import conformer_oracle as C

    ref_enc, ref_ctc, ref_dec, model, _ = _hybrid_setup(seed=33)
    vocab, b, lmax = 97, 3, 40
    g = torch.Generator().manual_seed(77)
    xs = torch.randn(b, 400, 80, generator=g)
    mask = torch.ones(b, 1, 400)
    mask[1, 0, 350:] = 0
    xs[1, 350:] = 0
    sub = C.subsample_mask(mask)
    ys_lens = torch.tensor([40, 33, 7], dtype=torch.int32)
    ys = torch.full((b, lmax), -1, dtype=torch.int32)
    eos = vocab - 1
    ys_in = torch.full((b, lmax + 1), eos, dtype=torch.int32)
    ys_out = torch.full((b, lmax + 1), -1, dtype=torch.int32)
    ys_masks = torch.zeros(b, 1, lmax + 1)
    for i, n in enumerate(ys_lens.tolist()):
        ys[i, :n] = torch.randint(1, vocab - 1, (n,), generator=g, dtype=torch.int32)
        ys_in[i, 1:n + 1] = ys[i, :n]
        ys_out[i, :n] = ys[i, :n]
        ys_out[i, n] = eos
        ys_masks[i, 0, :n + 1] = 1
    ys_sub = (ys_masks.bool() & torch.tril(torch.ones(lmax + 1, lmax + 1, dtype=torch.bool))[None]).float()
    cols = (xs, ys, ys_in, ys_out, None, None, sub, ys_sub, ys_masks, ys_lens, None)
    loss_ref, acc_ref, lc_ref, la_ref = C.hybrid_loss(ref_enc, ref_ctc, ref_dec, cols, 0.3, 0.1, False)
    loss_ref.backward()
    eng = ConformerCTCTrainStep(model, dropout_rate=0.0, positional_dropout_rate=0.0,
                                compute_type=torch.float32 if mode == "float32" else None)
    dev = [c.cuda() if c is not None else None for c in cols]
    loss = eng.forward_backward(dev[0], dev[1], dev[6], dev[9], None, 1.0, ys_in_pad=dev[2], ys_out_pad=dev[3], ys_sub_masks=dev[7],
                                ys_masks=dev[8])
    tol = 2e-5 if mode == "float32" else 2e-2
    assert abs(float(eng.last_loss_att) - float(la_ref.detach())) <= tol * abs(float(la_ref.detach()))
    assert abs(float(loss) - float(loss_ref.detach())) <= tol * abs(float(loss_ref.detach()))
    grads = eng.gradients()
    want = {"encoder." + n: p.grad for n, p in ref_enc.named_parameters()}
    want.update({"ctc." + n: p.grad for n, p in ref_ctc.named_parameters()})
    want.update({"decoder." + n: p.grad for n, p in ref_dec.named_parameters()})
    lim = 5e-4 if mode == "float32" else 6e-2
    bad = {k: round(rel_rms(grads[k], w), 5) for k, w in want.items()
           if "depthwise_conv.bias" not in k and "linear_k.bias" not in k and rel_rms(grads[k], w) > lim}
    assert not bad, bad
    out = eng.step(*dev)  # ... and the optimizer step runs end to end on it (launch table recorded on a later sighting)
    assert not out[1] and float(out[0]) > 0


def test_adam_bias_correction_counts_applied_updates_only():
    """After a skipped (overflow) step the next applied update must equal what Adam does at its own step count: the engine with
    an overflow step in the middle ends on the same masters as one that never saw the bad batch, when the LR is constant across
    steps (warm-up long past: base_lr * sqrt(w) * s^-0.5 is replaced here by a flat schedule through warmup_steps=1, lr ~ s^-0.5
    differs per step, so compare against an explicit float64 Adam on the recorded gradients instead)."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep, asr_warmup_lr

    _, _, model = build(blocks=1, seed=8, cmvn=False)
    xs, ys, sub, ys_lens = batch(b=2, seed=13)
    eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=2, dropout_rate=0.0, positional_dropout_rate=0.0)
    cols = (xs.cuda(), ys.cuda(), None, None, None, None, sub.cuda(), None, None, ys_lens.cuda(), None)
    xs_bad = xs.clone()
    xs_bad[0, 3, 5] = float("inf")
    name = "after_norm.g"
    p = eng.fp.p(name).double().cpu().clone()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    applied = 0
    for call, bad in enumerate([False, True, False, False]):
        out = eng.step(xs_bad.cuda() if bad else cols[0], *cols[1:])
        assert bool(out[3]) == bad
        if bad:
            continue
        g = eng.fp.g(name).double().cpu() / out[2]       # unscaled gradient of this step
        applied += 1
        lr = asr_warmup_lr(call, 1e-3, 2)                # LR index = number of get_lr() calls so far = step() calls
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        lr_t = lr * math.sqrt(1 - 0.999 ** applied) / (1 - 0.9 ** applied)
        p = p - lr_t * m / (v.sqrt() + 1e-8)
    assert torch.allclose(eng.fp.p(name).double().cpu(), p, rtol=0, atol=2e-6)


def test_lr_step_rule_mindspore23_advances_twice_per_applied_step():
    from mindaudio_amd.train.engine import ConformerCTCTrainStep, asr_warmup_lr

    _, _, model = build(blocks=1, seed=8, cmvn=False)
    xs, ys, sub, ys_lens = batch(b=2, seed=13)
    eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=50, dropout_rate=0.0, positional_dropout_rate=0.0,
                                lr_step_rule="mindspore23")
    cols = (xs.cuda(), ys.cuda(), None, None, None, None, sub.cuda(), None, None, ys_lens.cuda(), None)
    lrs = [eng.step(*cols)[4] for _ in range(3)]
    assert lrs == [asr_warmup_lr(s, 1e-3, 50) for s in (0, 2, 4)] and eng.global_step == 6 and eng.applied_steps == 3


def test_fused_blocks_equal_the_one_launch_per_cell_path_with_dropout():
    """The fused block launches (packed dense layers with Swish / dropout / residual / LayerNorm / Swish' / the next branch's dropout
    backward in their epilogues) against the one-launch-per-cell path of the same engine, dropout ON: the counter-based masks are a
    function of (seed, site, element index), so both paths drop the same elements; what differs is float32 summation order inside the
    LayerNorms and the float32 (instead of bf16-rounded) join input.  Loss and every gradient agree to bf16 round-off."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    xs, ys, sub, ys_lens = batch()
    res = []
    for fused in (False, True):
        _, _, model = build(seed=6)
        eng = ConformerCTCTrainStep(model, dropout_rate=0.1, positional_dropout_rate=0.1, fused=fused)
        assert eng.fused == fused
        loss = eng.forward_backward(xs.cuda(), ys.cuda(), sub.cuda(), ys_lens.cuda(), grad_scale=8.0)
        res.append((float(loss), eng.fp.grad.clone(), eng))
    (l0, g0, e0), (l1, g1, e1) = res
    assert abs(l0 - l1) <= 2e-3 * abs(l0), (l0, l1)
    assert float((g1 - g0).norm() / g0.norm()) <= 2e-2
    worst = {}
    for name, (off, shape, n) in e0.fp.index.items():
        a, b_ = g0[off:off + n], g1[off:off + n]
        if float(a.norm()) > 1e-6 * float(g0.norm()):
            worst[name] = float((a - b_).norm() / a.norm())
    bad = {k: round(v, 4) for k, v in worst.items() if v > 6e-2 and not (k.endswith("dw_b") or k.endswith("qkv_b"))}
    assert not bad, bad
    # and the fused path is run-to-run bit-reproducible
    _, _, model = build(seed=6)
    eng = ConformerCTCTrainStep(model, dropout_rate=0.1, positional_dropout_rate=0.1, fused=True)
    loss = eng.forward_backward(xs.cuda(), ys.cuda(), sub.cuda(), ys_lens.cuda(), grad_scale=8.0)
    assert float(loss) == l1 and torch.equal(eng.fp.grad, g1)


def test_feed_forward_module_in_one_launch_changes_round_off_only():
    """ffn_one_launch (ma_ffn_train_bf16: w_1 + Swish + dropout + w_2 + join in one launch, the tape stored from the registers that feed
    the second product) and ffn_bwd_one_launch (ma_ffn_train_bwd_bf16: dh -> du -> da -> LayerNorm backward in one launch, on the
    gk = swish' * keep / (1 - p) tape) against the two-launch forms of the same fused engine, dropout ON: the same masks; u may differ
    by one bf16 ulp, h by the rounding of Swish's argument, du by gk's rounding.  Loss and gradients agree to bf16 round-off, and
    the one-launch path is run-to-run bit-reproducible."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    xs, ys, sub, ys_lens = batch()
    res = []
    for fwd_one, bwd_one, chained in ((False, False, False), (True, False, False), (True, True, False), (True, True, True), (True, True, True)):
        _, _, model = build(seed=6)
        eng = ConformerCTCTrainStep(model, dropout_rate=0.1, positional_dropout_rate=0.1, fused=True)
        assert eng.ffn_one_launch and eng.ffn_bwd_one_launch and eng.ln_final_chained
        eng.ln_final_chained = chained  # norm_final's backward as the second stage of the macaron backward launch above it
        if not (fwd_one and bwd_one):
            eng.ffn_one_launch, eng.ffn_bwd_one_launch = fwd_one, bwd_one
            eng._pack_plan = None
            eng._pack_weights()
        loss = eng.forward_backward(xs.cuda(), ys.cuda(), sub.cuda(), ys_lens.cuda(), grad_scale=8.0)
        res.append((float(loss), eng.fp.grad.clone(), eng))
    (l0, g0, e0) = res[0]
    for l1, g1, _ in res[1:4]:
        assert abs(l0 - l1) <= 1e-3 * abs(l0), (l0, l1)
        assert float((g1 - g0).norm() / g0.norm()) <= 1.5e-2
        worst = {}
        for name, (off, shape, n) in e0.fp.index.items():
            a, b_ = g0[off:off + n], g1[off:off + n]
            if float(a.norm()) > 1e-6 * float(g0.norm()):
                worst[name] = float((a - b_).norm() / a.norm())
        bad = {k: round(v, 4) for k, v in worst.items() if v > 5e-2 and not (k.endswith("dw_b") or k.endswith("qkv_b"))}
        assert not bad, bad
    assert res[4][0] == res[3][0] and torch.equal(res[4][1], res[3][1])
    # the chained form against the un-chained one: the same launches but for norm_final's row sums taken in another order
    assert float((res[3][1] - res[2][1]).norm() / res[2][1].norm()) <= 1e-4


def test_blocks_issued_from_the_launch_table_change_no_bit():
    """block_tables (ma_conformer_block_fwd_train / _bwd_train, csrc/block_table.hip): the first step of a batch shape is walked from
    Python, the second is walked AND recorded, from the third on every block is ONE C call each way.  Same entry points, same
    arguments, same buffers, same order - twelve optimizer steps (dropout ON: the seed changes every step) give bit-identical losses,
    gradients and masters with the table on and off, through a second batch shape and the streaming configuration's chunk masks
    (the table follows the shape's plan and the mask's shape)."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    # A DIFFERENT batch of the same shape at every step (features, pad lengths, labels, label lengths, and the chunk size of the
    # streaming masks): a value derived from the masks or labels by an op the table does not re-issue would be stale on replay - with
    # one repeated batch the stale value would equal the fresh one and the comparison could not see it.
    def cols_of(k, form):
        xs, ys, sub, ys_lens = batch(seed=9 + k, shorter=(0, 30 - 2 * k, 61 + k), ylens=(9, 6 - k % 3, 4 + k % 4))
        if form == "short":
            t_short = xs.shape[1] - 16
            xs = xs[:, :t_short].contiguous()
            sub = sub[:, :, :((t_short - 3) // 2 + 1 - 3) // 2 + 1].contiguous()
        c = (xs.cuda(), ys.cuda(), None, None, None, None, sub.cuda(), None, None, ys_lens.cuda(), None)
        if form == "chunk":
            idx, cs = torch.arange(sub.shape[-1]), 8 - k % 3
            chunk = ((idx[None, :] // cs) <= (idx[:, None] // cs)) & ((idx[None, :] // cs) >= (idx[:, None] // cs) - 2)
            c = c[:10] + ((chunk[None] & (sub > 0)).float().cuda(),)  # the streaming configuration's masks (utils/mask.py:201-271)
        return c
    out = []
    for tables in (False, True):
        _, _, model = build(seed=9)
        eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=2, dropout_rate=0.1, positional_dropout_rate=0.1)
        assert eng.block_tables
        eng.block_tables = tables
        losses, states = [], []
        for k in range(12):
            losses.append(float(eng.step(*cols_of(k, "short" if k in (3, 4, 5) else "chunk" if k >= 9 else "plain"))[0]))
            tb = eng._dw_plan.get("table")
            states.append(None if tb is None else tb["state"])
        torch.cuda.synchronize()
        if tables:
            # shape A: seen, record, replay | shape B (a new plan): seen, record, replay | shape A again: its plan and table were kept
            # | shape A with (B, T', T') chunk masks (same plan, another table): seen, record, replay
            assert states == ["seen", "replay", "replay"] * 2 + ["replay"] * 3 + ["seen", "replay", "replay"], states
            assert len(eng._dw_plans) == 2 and len(eng._dw_plan["tables"]) == 2
            tab = eng._dw_plan["table"]["table"]
            per_block = [(tab.calls(False, li), tab.calls(True, li)) for li in range(eng.L)]
            assert all(f == 10 and bw >= 10 for f, bw in per_block), per_block
            assert tab.calls(False, eng.L) == 0
        else:
            assert states == [None] * 12
        out.append((losses, eng.fp.grad.clone(), eng.fp.master.clone()))
    assert out[1][0] == out[0][0]
    assert torch.equal(out[1][1], out[0][1]) and torch.equal(out[1][2], out[0][2])


def test_launch_table_budget_eviction_and_unrecordable_shapes_change_no_bit(monkeypatch):
    """The launch table's fallbacks: (a) a byte budget smaller than one table - the first shape's table stays, the second
    shape's is given back as soon as it is recorded and that shape is walked from then on (round 6: dropping the OLDER table made a
    loader's round-robin over its buckets evict, re-walk and re-record a shape on every step); (b) a recording that meets a call it cannot store - the shape is
    walked from Python from then on (state 'walk', one warning).  Losses and masters stay bit-identical to the engine without tables."""
    import warnings

    from mindaudio_amd import _lib
    from mindaudio_amd.train import block_table as BT
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    def cols_of(k, short):
        xs, ys, sub, ys_lens = batch(seed=20 + k, shorter=(0, 20 + k, 50 - k), ylens=(9, 5 + k % 3, 4))
        if short:
            t_short = xs.shape[1] - 16
            xs = xs[:, :t_short].contiguous()
            sub = sub[:, :, :((t_short - 3) // 2 + 1 - 3) // 2 + 1].contiguous()
        return (xs.cuda(), ys.cuda(), None, None, None, None, sub.cuda(), None, None, ys_lens.cuda(), None)

    order = [False, False, False, True, True, True, False, False, True, False]  # shape A x3, B x3, A, A, B, A
    out = []
    for mode in ("walked", "budget", "broken", "oom"):
        _, _, model = build(seed=9)
        eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=2, dropout_rate=0.1, positional_dropout_rate=0.1)
        eng.block_tables = mode != "walked"
        if mode == "budget":
            eng.block_table_max_bytes = 1
        if mode == "broken":
            real_add, calls = BT.BlockTable._add, [0]

            def flaky_add(self, name, *a):
                calls[0] += 1
                if calls[0] == 40:  # somewhere inside the first recorded step
                    raise _lib.MindaudioAmdError("injected: %s cannot be stored" % name)
                return real_add(self, name, *a)
            monkeypatch.setattr(BT.BlockTable, "_add", flaky_add)
        if mode == "oom":
            # (c) the device runs out of memory in the middle of a RECORDING step (after the forward pass has moved the BatchNorm running
            # statistics): tables dropped, statistics restored, the shape walked, the step run again
            real_bwd, fired = eng._blocks_backward_fused, [0]

            def oom_once(g, tape, dpos_all, c):
                tb = c.get("table")
                if tb is not None and tb["state"] == "record" and not fired[0]:
                    fired[0] = torch.cuda.memory_allocated()  # (what the failed attempt holds: its tape + the table's pinned buffers)
                    raise torch.OutOfMemoryError("injected")
                return real_bwd(g, tape, dpos_all, c)
            eng._blocks_backward_fused = oom_once
        losses, states = [], []
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for k, short in enumerate(order):
                losses.append(float(eng.step(*cols_of(k, short))[0]))
                tb = eng._dw_plan.get("table")
                states.append(None if tb is None else tb["state"])
        torch.cuda.synchronize()
        if mode == "budget":
            # shape A is recorded and replays to the end; shape B's recording goes over the budget: given back, walked for good
            assert states[1] == "replay" and states[4] == "walk" and states[5] == "walk" and states[9] == "replay", states
            assert len(eng._table_bytes) == 1 and any("block_table_max_bytes" in str(w.message) for w in caught)
        if mode == "broken":
            monkeypatch.setattr(BT.BlockTable, "_add", real_add)
            assert states[1] == "walk" and states[2] == "walk" and states[4] == "replay", states  # shape A walked for good, shape B recorded
            assert any("walked from Python" in str(w.message) for w in caught)
        if mode == "oom":
            assert fired[0] > 0 and states[1] == "walk" and states[2] == "walk" and states[4] == "replay", states
            assert any("out of memory while recording" in str(w.message) for w in caught)
            # the retry ran with the failed attempt's memory GIVEN BACK (a retry inside the except block keeps the failed
            # frames - tape, activations, the recording table's buffers - alive through the traceback)
            assert eng._oom_freed_to < fired[0] - (8 << 20), (eng._oom_freed_to, fired[0])
        out.append((losses, eng.fp.master.clone(), [m_.clone() for m_ in eng.bn_mean]))
    for got in out[1:]:
        assert got[0] == out[0][0] and torch.equal(got[1], out[0][1])
        assert all(torch.equal(a_, b_) for a_, b_ in zip(got[2], out[0][2]))  # BatchNorm running statistics too


def test_hybrid_step_from_the_launch_table_changes_no_bit():
    """The shipped configuration (ctc_weight 0.3): the decoder's layers join the encoder blocks' launch table (decoder layer l =
    block L + l; embedding, output layer and its backward = blocks L + Ld .. L + Ld + 2), the label-smoothing loss - which takes the
    step's loss scale as an argument - stays a live call between the replayed segments.  Seven optimizer steps, dropout ON in encoder
    and decoder, the loss scale changed by hand after the fourth: bit-identical losses and masters with the table on and off."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    out = []
    for tables in (False, True):
        _, _, _, model, cols = _hybrid_setup(blocks=2, dblocks=2)
        model.decoder.dropout_rate, model.decoder.positional_dropout_rate = 0.1, 0.1
        eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=2, dropout_rate=0.1, positional_dropout_rate=0.1)
        assert eng.block_tables and eng.dec is not None and float(eng.dec.dropout_rate) == 0.1
        eng.block_tables = tables
        losses = []
        for k in range(7):
            if k == 4:
                eng.scaler.scale = 256.0
            # (another batch of the same shape every step: see test_blocks_issued_from_the_launch_table_change_no_bit)
            step_cols = _hybrid_cols(96, batch(vocab=96, seed=12 + k, shorter=(0, 30 - k, 61 + 2 * k), ylens=(9, 6 - k % 2, 4 + k % 3)))
            losses.append(float(eng.step(*(c.cuda() if c is not None else None for c in step_cols))[0]))
        torch.cuda.synchronize()
        if tables:
            tb = eng._dw_plan["table"]
            tab, L, Ld = tb["table"], eng.L, eng.Ld
            assert tb["state"] == "replay" and "out" in tb["dec"]
            counts = [(tab.calls(False, blk), tab.calls(True, blk)) for blk in range(L, L + Ld + 3)]
            assert all(f > 0 and bw > 0 for f, bw in counts[:Ld + 1]) and counts[Ld + 1][0] > 0 and counts[Ld + 2][0] > 0, counts
        out.append((losses, eng.fp.master.clone()))
    assert out[1][0] == out[0][0]
    assert torch.equal(out[1][1], out[0][1])


def test_hybrid_table_with_label_lengths_that_change_from_batch_to_batch():
    """Real batches differ in their longest transcript: the decoder's table replays only when the label width is the recorded one,
    otherwise the decoder is walked from Python WHILE the encoder's backward is replayed - whose first recorded weight-gradient group
    holds the decoder's memory-side product (ca_kv).  Until the end of round 6 that combination read another step's operand (a wrong
    gradient) and left a dangling item behind (a memory fault a few steps later; found by tools/hybrid_soak.py).  Twelve steps over
    label widths 10 / 7 / 13 on one encoder shape: bit-identical losses and masters with the tables on and off."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    def cols_for(k):
        lmax = (9, 9, 9, 6, 9, 12, 6, 9, 12, 12, 9, 6)[k]
        xs, ys, sub, ys_lens = batch(vocab=96, seed=40 + k, shorter=(0, 30 - k, 61 + 2 * k), ylens=(min(9, lmax), 6 - k % 2, 4 + k % 3))
        b, eos = ys.shape[0], 96
        ys_in = torch.full((b, lmax + 1), eos, dtype=torch.int32)
        ys_out = torch.full((b, lmax + 1), -1, dtype=torch.int32)
        ys_masks = torch.zeros(b, 1, lmax + 1)
        ys_w = torch.full((b, lmax), -1, dtype=torch.int32)
        for i, n in enumerate(ys_lens.tolist()):
            ys_w[i, :n] = ys[i, :n]
            ys_in[i, 1:n + 1] = ys[i, :n]
            ys_out[i, :n] = ys[i, :n]
            ys_out[i, n] = eos
            ys_masks[i, 0, :n + 1] = 1
        ys_sub = (ys_masks.bool() & torch.tril(torch.ones(lmax + 1, lmax + 1, dtype=torch.bool))[None]).float()
        return (xs, ys_w, ys_in, ys_out, None, None, sub, ys_sub, ys_masks, ys_lens, None)

    out = []
    for tables in (False, True):
        _, _, _, model, _ = _hybrid_setup(blocks=2, dblocks=2)
        model.decoder.dropout_rate, model.decoder.positional_dropout_rate = 0.1, 0.1
        eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=2, dropout_rate=0.1, positional_dropout_rate=0.1)
        eng.block_tables = tables
        losses = []
        for k in range(12):
            losses.append(float(eng.step(*(c.cuda() if c is not None else None for c in cols_for(k)))[0]))
        torch.cuda.synchronize()
        if tables:
            tb = eng._dw_plan["table"]
            assert tb["state"] == "replay" and tb["dec"]["L1"] == 10  # (recorded at the second sighting: width 10)
            assert eng._dq is None or eng._dq.n == 0  # nothing queued that no launch will take
        out.append((losses, eng.fp.master.clone()))
    assert out[1][0] == out[0][0]
    assert torch.equal(out[1][1], out[0][1])


def test_no_second_stream_option_and_full_direct_groups_flush():
    """The step runs on one stream: the rounds-3/4 constructor options that put weight-gradient work on a second stream are gone
    (VERDICT r4 #6c; the unexplained two-queue corruption of DESIGN 4.6.3 is reachable only through the reproducer hook of tools/).
    And a direct weight-gradient group that outgrows the kernel's item table leaves as two grids instead of raising:
    same gradients as with room to spare."""
    from mindaudio_amd import _lib
    from mindaudio_amd.train import kernels as K
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    for kw in ("wg_stream", "_split_k_sums_on_second_stream"):
        with pytest.raises(TypeError):
            ConformerCTCTrainStep(build(seed=9)[2], **{kw: True})
    eng = ConformerCTCTrainStep(build(seed=9)[2], dw_group_blocks=100)
    cap = int(_lib.load().ma_gemm_tn_direct_max_items())
    assert eng._wg_stream is None and 1 <= eng.dw_group_blocks <= cap // 8
    # cap + 3 products through one group: the first `cap` leave when the table is full
    g = torch.Generator(device="cpu").manual_seed(5)
    dys = [torch.randn(512, 256, generator=g).bfloat16().cuda() for _ in range(cap + 3)]
    xs_ = [torch.randn(512, 256, generator=g).bfloat16().cuda() for _ in range(cap + 3)]
    outs = [torch.zeros(256, 256, device="cuda") for _ in range(cap + 3)]
    sums = [torch.zeros(256, device="cuda") for _ in range(cap + 3)]
    grp = K.DirectGroup()
    if not K.gemm_tn_direct_ok(dys[0], xs_[0], outs[0]):
        pytest.skip("direct products not available for this shape")
    for dy, x, o, c_ in zip(dys, xs_, outs, sums):
        grp.add(dy, x, o, c_)
    assert grp.n == 3
    grp.launch()
    grp.clear()
    torch.cuda.synchronize()
    for dy, x, o, c_ in zip(dys, xs_, outs, sums):
        want = dy.double().T @ x.double()
        assert float((o.double() - want).abs().max()) <= 2e-4 * float(want.abs().max()) + 1e-3
        assert float((c_.double() - dy.double().sum(0)).abs().max()) <= 1e-2


def test_fused_engine_gradients_match_oracle_autograd():
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    ref_enc, ref_ctc, model = build()
    xs, ys, sub, ys_lens = batch()
    loss_ref = oracle_loss(ref_enc, ref_ctc, xs, ys, sub, ys_lens)
    loss_ref.backward()
    eng = ConformerCTCTrainStep(model, dropout_rate=0.0, positional_dropout_rate=0.0, fused=True)
    loss = eng.forward_backward(xs.cuda(), ys.cuda(), sub.cuda(), ys_lens.cuda(), grad_scale=1.0)
    assert abs(float(loss) - float(loss_ref.detach())) <= 2e-2 * abs(float(loss_ref.detach()))
    grads = eng.gradients()
    want = {"encoder." + n: p.grad for n, p in ref_enc.named_parameters()}
    want.update({"ctc." + n: p.grad for n, p in ref_ctc.named_parameters()})
    worst = {n: rel_rms(grads[n], gw) for n, gw in want.items() if "depthwise_conv.bias" not in n and "linear_k.bias" not in n}
    bad = {k: round(v, 4) for k, v in worst.items() if v > 6e-2}
    assert not bad, bad
    assert sum(worst.values()) / len(worst) < 2.5e-2


def test_evaluation_between_training_steps_sees_the_trained_decoder():
    """train.py's EvalCallback evaluates the module between training steps: sync_to_module() has to invalidate EVERY packed copy the
    evaluation forward keeps - the encoder's, the CTC head's and the decoder's.  Until the end of round 6 the decoder's stayed: the
    evaluation loss of a hybrid model stopped at the untrained decoder's (found by tools/recipe_learns.py --with-eval).  Here: evaluate
    (fills the caches), train five steps, sync, evaluate again - equal to a fresh module loaded with the same state, and moved."""
    import copy

    from mindaudio_amd.conformer.asr_model import ASREvalNet
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    _, _, _, model, cols = _hybrid_setup(blocks=1, dblocks=1)
    dev_cols = tuple(c.cuda() if c is not None else None for c in cols)
    model.eval()
    before = float(ASREvalNet(model, 1)(*dev_cols))
    eng = ConformerCTCTrainStep(model, base_lr=2e-3, warmup_steps=1, dropout_rate=0.0, positional_dropout_rate=0.0)
    for _ in range(6):
        eng.step(*dev_cols)
    eng.sync_to_module()
    model.eval()
    after = float(ASREvalNet(model, 1)(*dev_cols))
    fresh = copy.deepcopy(model)
    fresh.load_state_dict(model.state_dict())  # (the load hook drops every packed copy)
    fresh.eval()
    want = float(ASREvalNet(fresh, 1)(*dev_cols))
    assert after == want, (before, after, want)
    assert after < 0.9 * before, (before, after)
    # and the decoder alone: with the encoder / CTC caches valid and only the decoder's stale the loss would sit between the two
    dec_w = model.decoder.decoders[0].feed_forward.w_1.weight
    assert not torch.equal(dec_w, copy.deepcopy(_hybrid_setup(blocks=1, dblocks=1)[3]).decoder.decoders[0].feed_forward.w_1.weight)


def test_weights_loaded_into_the_module_behind_the_engine_reach_it_through_sync_from_module():
    """The constructor reads the module once.  sync_from_module() makes a later load_state_dict the engine's masters (and BatchNorm
    statistics, and fresh Adam moments): three steps after it equal, bit for bit, those of an engine built on the loaded module."""
    import copy

    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    _, _, _, model_a, cols = _hybrid_setup(seed=31, blocks=1, dblocks=1)
    _, _, _, model_b, _ = _hybrid_setup(seed=32, blocks=1, dblocks=1)
    with torch.no_grad():
        for l in model_b.encoder.encoders:
            l.conv_module.norm.running_mean.normal_(0, 0.1)
            l.conv_module.norm.running_var.uniform_(0.5, 1.5)
    state_b = copy.deepcopy(model_b.state_dict())
    dev_cols = tuple(c.cuda() if c is not None else None for c in cols)
    kw = dict(base_lr=1e-3, warmup_steps=1, dropout_rate=0.1, positional_dropout_rate=0.1)
    eng = ConformerCTCTrainStep(model_a, **kw)
    eng.step(*dev_cols)                       # (moments and counters move; the loaded weights must not inherit the moments)
    model_a.load_state_dict(state_b)
    eng.sync_from_module()
    eng.calls = eng.global_step = eng.applied_steps = 0   # same dropout stream and schedule position as the fresh engine
    ref = ConformerCTCTrainStep(model_b, **kw)
    got = [float(eng.step(*dev_cols)[0]) for _ in range(3)]
    want = [float(ref.step(*dev_cols)[0]) for _ in range(3)]
    assert got == want and torch.equal(eng.fp.master, ref.fp.master)
    assert all(torch.equal(a, b) for a, b in zip(eng.bn_mean + eng.bn_var, ref.bn_mean + ref.bn_var))


def _hybrid_setup(seed=31, vocab=97, blocks=1, dblocks=2):
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from oracle import conformer_oracle as C

    torch.manual_seed(seed)
    ref_enc = C.ConformerEncoder(80, 256, 4, 2048, blocks, dropout_rate=0.0, positional_dropout_rate=0.0).train()
    ref_ctc = C.CTC(vocab, 256).train()
    ref_dec = C.TransformerDecoder(vocab, 256, 4, 512, dblocks, 0.0, 0.0).train()
    with torch.no_grad():
        for mod in list(ref_enc.modules()) + list(ref_dec.modules()):
            if isinstance(mod, C.LayerNorm):
                mod.gamma.uniform_(0.8, 1.2)
                mod.beta.normal_(0, 0.1)
    model = create_asr_model(80, vocab, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=blocks),
                             ctc_weight=0.3, decoder_conf=dict(attention_heads=4, linear_units=512, num_blocks=dblocks,
                                                               dropout_rate=0.0, positional_dropout_rate=0.0), lsm_weight=0.1)
    model.encoder.load_state_dict(ref_enc.state_dict(), strict=False)
    model.ctc.load_state_dict(ref_ctc.state_dict())
    model.decoder.load_state_dict(ref_dec.state_dict(), strict=False)
    return ref_enc, ref_ctc, ref_dec, model.cuda(), _hybrid_cols(vocab - 1, batch(vocab=vocab - 1, seed=12))


def _hybrid_cols(eos, b4):
    xs, ys, sub, ys_lens = b4
    b, lmax = ys.shape[0], 9
    sos = eos
    ys_in = torch.full((b, lmax + 1), eos, dtype=torch.int32)
    ys_out = torch.full((b, lmax + 1), -1, dtype=torch.int32)
    ys_masks = torch.zeros(b, 1, lmax + 1)
    for i, n in enumerate(ys_lens.tolist()):
        ys_in[i, 0] = sos
        ys_in[i, 1:n + 1] = ys[i, :n]
        ys_out[i, :n] = ys[i, :n]
        ys_out[i, n] = eos
        ys_masks[i, 0, :n + 1] = 1
    ys_sub = (ys_masks.bool() & torch.tril(torch.ones(lmax + 1, lmax + 1, dtype=torch.bool))[None]).float()
    return (xs, ys, ys_in, ys_out, None, None, sub, ys_sub, ys_masks, ys_lens, None)


def test_decoder_embedding_backward_skips_padded_rows_without_changing_a_bit():
    """ma_embed_bwd_rows_f32 (round 6): the padded label positions (a third of the B x L rows, all the <eos> token) are skipped in the
    embedding's backward.  That is exact, not approximate: the (B, L, L) label mask keeps them out of every valid position's
    attention and the loss ignores them, so their rows of the gradient are zero - the whole flat gradient must be bit-identical with
    and without the skip, dropout on, and the kernel must agree with the un-masked entry point on a gradient whose padded rows are
    zero."""
    from mindaudio_amd.train import kernels as K
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    grads = []
    for skip in (True, False):
        _, _, _, model, cols = _hybrid_setup(blocks=1, dblocks=2)
        model.decoder.dropout_rate, model.decoder.positional_dropout_rate = 0.1, 0.1
        eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=2, dropout_rate=0.1, positional_dropout_rate=0.1)
        assert eng.decoder_embed_row_mask
        eng.decoder_embed_row_mask = skip
        c = [x.cuda() if x is not None else None for x in cols]
        eng.forward_backward(c[0], c[1], c[6], c[9], grad_scale=64.0, ys_in_pad=c[2], ys_out_pad=c[3], ys_sub_masks=c[7], ys_masks=c[8])
        torch.cuda.synchronize()
        grads.append(eng.fp.grad.clone())
        assert float(eng.fp.g("dec.embed").abs().max()) > 0
    assert torch.equal(grads[0], grads[1])
    # the entry point alone: rows with row_keep == 0 contribute nothing; with zero gradient there it equals the plain entry point
    g = torch.Generator().manual_seed(3)
    rows, d, v = 1240, 256, 97
    tok = torch.randint(0, v, (rows,), generator=g, dtype=torch.int32)
    keep = (torch.rand(rows, generator=g) > 0.4).float()
    tok[keep == 0] = v - 1
    gr = torch.randn(rows, d, generator=g) * keep[:, None]
    a, b_ = torch.zeros(v, d, device="cuda"), torch.zeros(v, d, device="cuda")
    K.embed_bwd(tok.cuda(), gr.cuda(), a, 16.0, 0.1, 5, 7)
    K.embed_bwd(tok.cuda(), gr.cuda(), b_, 16.0, 0.1, 5, 7, row_keep=keep.cuda())
    assert torch.equal(a, b_)
    c_ = torch.zeros(v, d, device="cuda")
    K.embed_bwd(tok.cuda(), torch.randn(rows, d, generator=g).cuda(), c_, 16.0, 0.0, 5, 7, row_keep=torch.zeros(rows, device="cuda"))
    assert float(c_.abs().max()) == 0.0


def test_hybrid_loss_curve_in_float32_mode_matches_the_oracle():
    """The shipped conformer.yaml trains with ctc_weight 0.3 (asr_model.py:117-153): the float32 validation mode now covers the
    attention-decoder branch too (float32 decoder attention, embedding, label smoothing).  20 Adam steps on one batch against the
    same recipe on the float32 PyTorch oracle: every loss within 1e-4 (north star), first-step gradients within 2e-4."""
    from mindaudio_amd.train.engine import ConformerCTCTrainStep, asr_warmup_lr
    from oracle import conformer_oracle as C

    ref_enc, ref_ctc, ref_dec, model, cols = _hybrid_setup()
    params = list(ref_enc.parameters()) + list(ref_ctc.parameters()) + list(ref_dec.parameters())
    opt = torch.optim.Adam(params, lr=1.0, betas=(0.9, 0.999), eps=1e-8)
    warm, base, n_steps = 25, 2e-4, 20
    eng = ConformerCTCTrainStep(model, base_lr=base, warmup_steps=warm, dropout_rate=0.0, positional_dropout_rate=0.0,
                                compute_type=torch.float32)
    dev = [c.cuda() if c is not None else None for c in cols]
    got, want = [], []
    for step in range(n_steps):
        for gq in opt.param_groups:
            gq["lr"] = asr_warmup_lr(step, base, warm)
        opt.zero_grad()
        l_ref, _, _, _ = C.hybrid_loss(ref_enc, ref_ctc, ref_dec, cols, 0.3, 0.1)
        l_ref.backward()
        if step == 0:
            want_g = {"encoder." + n: p.grad.clone() for n, p in ref_enc.named_parameters()}
            want_g.update({"ctc." + n: p.grad.clone() for n, p in ref_ctc.named_parameters()})
            want_g.update({"decoder." + n: p.grad.clone() for n, p in ref_dec.named_parameters()})
        opt.step()
        want.append(float(l_ref.detach()))
        loss, cond, scale, overflow, _ = eng.step(*dev)
        assert not overflow
        got.append(float(loss))
        if step == 0:
            grads = eng.gradients()
            gmax = float(max(p.abs().max() for p in want_g.values()))
            for name, gw in want_g.items():
                if "depthwise_conv.bias" in name or "linear_k.bias" in name:
                    assert float(grads[name].abs().max()) / scale < 1e-3 * gmax
                    continue
                assert rel_rms(grads[name] / scale, gw) <= 2e-4, name
    dev_max = max(abs(a - b_) / abs(b_) for a, b_ in zip(got, want))
    print("hybrid float32 loss curve over %d steps: %.3f -> %.3f, max relative deviation %.2e" % (n_steps, want[0], want[-1], dev_max))
    assert want[-1] < want[0] and dev_max <= 1e-4, (got, want)
