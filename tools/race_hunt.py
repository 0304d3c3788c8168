"""Development: run the cfg-4 step repeatedly and report any spurious overflow (which gradient entries are not finite).
python tools/race_hunt.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from mindaudio_amd.conformer.asr_model import create_asr_model
    from mindaudio_amd.train import engine as _engine
    from mindaudio_amd.train.engine import ConformerCTCTrainStep

    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    wg = os.environ.get("WG", "1") == "1"
    dev = torch.device("cuda", 0)
    torch.manual_seed(777)
    model = create_asr_model(80, 4233, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12)).to(dev)
    _engine._TWO_QUEUE_REPRODUCER.update(wg_stream=bool(wg))  # (reproducer hook: not a constructor option)
    eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=4, dropout_rate=0.1, positional_dropout_rate=0.1)
    rng = np.random.RandomState(43)
    b, t = 40, 1024
    xs = torch.from_numpy(rng.randn(b, t, 80).astype(np.float32)).to(dev)
    t2 = ((t - 3) // 2 + 1 - 3) // 2 + 1
    masks = torch.ones(b, 1, t2, device=dev)
    ylens = rng.randint(5, 31, b).astype(np.int32)
    ys = np.full((b, 30), -1, np.int32)
    for i, n in enumerate(ylens):
        ys[i, :n] = rng.randint(1, 4232, n)
    cols = (xs, torch.from_numpy(ys).to(dev), None, None, None, None, masks, None, None, torch.from_numpy(ylens).to(dev), None)
    if os.environ.get("TESTBATCH", "0") == "1":  # the ragged batch of tests/test_cfg4_full_shape_gpu.py
        import importlib.util

        spec = importlib.util.spec_from_file_location("cfg4", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                         "tests", "test_cfg4_full_shape_gpu.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        xs_, ys_, sub_, yl_ = mod._batch(40, 43)
        cols = (xs_.to(dev), ys_.to(dev), None, None, None, None, sub_.to(dev), None, None, yl_.to(dev), None)
    bad = 0
    fresh = int(os.environ.get("FRESH", "0"))
    for s in range(steps):
        if fresh and s % fresh == 0:
            torch.manual_seed(777)
            model = create_asr_model(80, 4233, dict(output_size=256, attention_heads=4, linear_units=2048, num_blocks=12)).to(dev)
            _engine._TWO_QUEUE_REPRODUCER.update(wg_stream=bool(wg))  # (reproducer hook: not a constructor option)
            eng = ConformerCTCTrainStep(model, base_lr=1e-3, warmup_steps=4, dropout_rate=0.1, positional_dropout_rate=0.1)
            junk = torch.full((1 << 28,), float("nan"), device=dev)  # poison freed memory: the next empty() gets NaNs
            small = [torch.full((1 << 17,), float("nan"), device=dev) for _ in range(512)]  # ... in the small-block pool too
            small2 = [torch.full((1 << 10,), float("nan"), device=dev) for _ in range(4096)]
            del junk, small, small2
        loss, cond, scale, overflow, lr = eng.step(*cols)
        tail = eng.fp._grad_alloc[eng.fp.size + 1:]
        if bool((tail != 0).any()):
            print("step %d: the padding behind the flag was written: %s" % (s, tail[:8].tolist()), flush=True)
        if overflow or not np.isfinite(float(loss)):
            bad += 1
            g = eng.fp.grad
            nf = (~torch.isfinite(g)).nonzero().flatten()
            names = []
            for name, (off, shape, n) in eng.fp.index.items():
                k = int(((nf >= off) & (nf < off + n)).sum())
                if k:
                    names.append((name, k))
            print("step %d: overflow=%s loss=%s scale=%s non-finite entries %d in %s; flag=%d" %
                  (s, overflow, float(loss), scale, nf.numel(), names[:8], int(eng.flag.item())), flush=True)
    print("done: %d steps, %d bad, wg_stream=%s" % (steps, bad, wg))


main()
