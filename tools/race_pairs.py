"""Development: many fresh engines in one process, same seed and batch: every engine must reproduce the first one's loss curve and
masters bit for bit.  python tools/race_pairs.py [engines] [steps]   (WG=0: without the weight-gradient stream)"""
import hashlib
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train import engine as _engine
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    n_eng = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    wg = os.environ.get("WG", "1") == "1"
    fused = os.environ.get("FUSED", "1") == "1"
    dev = torch.device("cuda", 0)
    spec = importlib.util.spec_from_file_location("cfg4", os.path.join(ROOT, "tests", "test_cfg4_full_shape_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    xs_, ys_, sub_, yl_ = mod._batch(40, 43)
    cols = (xs_.to(dev), ys_.to(dev), None, None, None, None, sub_.to(dev), None, None, yl_.to(dev), None)
    conf = dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12)
    ref, bad = None, 0
    for k in range(n_eng):
        torch.manual_seed(777)
        _engine._TWO_QUEUE_REPRODUCER.update(wg_stream=bool(wg))  # (reproducer hook: not a constructor option)
        eng = ConformerCTCTrainStep(create_asr_model(80, 4233, conf).to(dev), base_lr=1e-3, warmup_steps=4, dropout_rate=0.1,
                                    positional_dropout_rate=0.1, fused=fused)
        losses, grads = [], []
        for _ in range(steps):
            loss, cond, scale, overflow, lr = eng.step(*cols)
            losses.append((float(loss), bool(overflow)))
            grads.append(hashlib.sha1(eng.fp.grad.cpu().numpy().tobytes()).hexdigest()[:12])
        sig = (losses, grads, hashlib.sha1(eng.fp.master.cpu().numpy().tobytes()).hexdigest()[:12])
        if ref is None:
            ref = sig
            ref_grads = None
        elif sig != ref:
            bad += 1
            first = next(i for i in range(steps) if grads[i] != ref[1][i] or losses[i] != ref[0][i])
            print("engine %d differs from engine 0 first at step %d: loss %s vs %s, grad hash %s vs %s" %
                  (k, first, losses[first], ref[0][first], grads[first], ref[1][first]), flush=True)
    print("done: %d engines x %d steps, %d differ; wg_stream=%s fused=%s" % (n_eng, steps, bad, wg, fused))


main()
