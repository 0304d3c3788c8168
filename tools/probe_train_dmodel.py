#!/usr/bin/env python3
"""Does the training engine run at d_model != 256?  Builds the oracle and the model at (d, heads), one block, and compares loss and
gradients of the one-launch-per-cell engine (fused=False) in bf16 and float32 modes with the oracle's autograd.
    python tools/probe_train_dmodel.py --d 512 --heads 8"""
import argparse
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--units", type=int, default=2048)
    ap.add_argument("--blocks", type=int, default=1)
    a = ap.parse_args()
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train.engine import ConformerCTCTrainStep
    from oracle import conformer_oracle as C
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from test_train_step_gpu import batch, oracle_loss, rel_rms

    torch.manual_seed(5)
    vocab = 97
    ref_enc = C.ConformerEncoder(80, a.d, a.heads, a.units, a.blocks, dropout_rate=0.0, positional_dropout_rate=0.0)
    ref_ctc = C.CTC(vocab, a.d)
    model = create_asr_model(80, vocab, dict(output_size=a.d, attention_heads=a.heads, linear_units=a.units, num_blocks=a.blocks))
    model.encoder.load_state_dict(ref_enc.state_dict(), strict=False)
    model.ctc.load_state_dict(ref_ctc.state_dict())
    model = model.cuda()
    ref_enc.train(), ref_ctc.train()
    xs, ys, sub, ys_lens = batch()
    loss_ref = oracle_loss(ref_enc, ref_ctc, xs, ys, sub, ys_lens)
    loss_ref.backward()
    want = {"encoder." + n: p.grad for n, p in ref_enc.named_parameters()}
    want.update({"ctc." + n: p.grad for n, p in ref_ctc.named_parameters()})
    for mode, kw in (("float32", dict(compute_type=torch.float32)), ("bf16 unfused", dict(fused=False)), ("bf16 fused", {})):
        try:
            eng = ConformerCTCTrainStep(model, dropout_rate=0.0, positional_dropout_rate=0.0, **kw)
            loss = eng.forward_backward(xs.cuda(), ys.cuda(), sub.cuda(), ys_lens.cuda(), grad_scale=1.0)
            torch.cuda.synchronize()
            grads = eng.gradients()
            worst = {n: rel_rms(grads[n], g) for n, g in want.items() if "depthwise_conv.bias" not in n and "linear_k.bias" not in n}
            top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
            print("%-13s loss %.5f (oracle %.5f)  worst gradients %s" % (mode, float(loss), float(loss_ref), [(k, "%.1e" % v) for k, v in top]))
        except Exception:
            print("%-13s FAILED" % mode)
            traceback.print_exc(limit=6)


if __name__ == "__main__":
    main()
