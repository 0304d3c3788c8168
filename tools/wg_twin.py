"""Development: the two-stream engine against a single-stream twin IN ONE PROCESS, a different batch and dropout seed every step,
the whole flat gradient compared bit for bit after every step (steady-state check of the cross-stream dependencies: a stale read
that fresh-process loops with one repeated batch cannot see).   python tools/wg_twin.py [steps] [mode]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train import engine as _engine
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    import wg_hunt as mod_

    mod = {"V": mod_.V, "batch": mod_.batch}
    dev = torch.device("cuda", 0)
    engs = []
    both_single = len(sys.argv) > 2 and sys.argv[2] == "single"  # tool check: two single-stream engines must agree
    for wg in ((False, False) if both_single else (True, False)):
        torch.manual_seed(777)
        model = create_asr_model(80, mod["V"], dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12)).to(dev)
        _engine._TWO_QUEUE_REPRODUCER.update(wg_stream=bool(wg))  # (reproducer hook: not a constructor option)
        e = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=4, dropout_rate=0.1, positional_dropout_rate=0.1)
        e._wg_from = 0
        engs.append(e)
    pool = [mod["batch"](40, 100 + i) for i in range(8)]  # eight different batches, cycled (host generation costs 60 ms each)
    pool = [(x.to(dev), y.to(dev), s.to(dev), l.to(dev)) for x, y, s, l in pool]
    bad = 0
    for s in range(steps):
        xs, ys, sub, yl = pool[s % len(pool)]
        cols = (xs, ys, None, None, None, None, sub, None, None, yl, None)
        r = [e.step(*cols) for e in engs]
        same = torch.equal(engs[0].fp.grad, engs[1].fp.grad) and float(r[0][0]) == float(r[1][0])
        if not same:
            bad += 1
            diff = (engs[0].fp.grad != engs[1].fp.grad).nonzero().flatten()
            names = []
            for name, (off, shape, n) in engs[0].fp.index.items():
                k = int(((diff >= off) & (diff < off + n)).sum())
                if k:
                    names.append((name, k))
            print("step %d: %d gradient entries differ; loss %r vs %r; %s" % (s, diff.numel(), float(r[0][0]), float(r[1][0]), names[:10]),
                  flush=True)
            # re-align the twin so that one event does not repeat itself for the rest of the run
            engs[0].fp.master.copy_(engs[1].fp.master)
            engs[0].fp.exp_avg.copy_(engs[1].fp.exp_avg)
            engs[0].fp.exp_avg_sq.copy_(engs[1].fp.exp_avg_sq)
            engs[0].refresh_weights()
    print("wg_twin: %d steps, %d mismatching steps" % (steps, bad))


main()
