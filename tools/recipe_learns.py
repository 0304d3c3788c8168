#!/usr/bin/env python3
"""Does the whole recipe LEARN?  A synthetic language a small model can pick up in a few hundred steps - every symbol a pair of tones
(0.16 s, a 40 ms gap between symbols), utterances of 3 ... 10 symbols - goes through the reference's recipe end to end:
`conformer.train` (wav files -> device Kaldi fbank -> SpecAugment -> buckets -> hybrid CTC / attention step with Adam, warm-up, loss
scale, BatchNorm statistics -> checkpoint in the reference's format) and `conformer.predict` (checkpoint -> greedy CTC -> CER) on
utterances the training never saw.  Prints the loss curve's ends and the held-out CER.
    python tools/recipe_learns.py [--epochs 40] [--blocks 2]"""
import argparse
import json
import os
import sys
import tempfile
import time
import wave

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

SYMBOLS = "abcdefghij"
LOW = (350.0, 480.0, 620.0, 790.0, 1000.0)      # symbol k = tone LOW[k % 5] + tone HIGH[k // 5]
HIGH = (1500.0, 2300.0)


def synth(rng, text, sr=16000):
    parts = [np.zeros(int(sr * rng.uniform(0.03, 0.08)))]
    for ch in text:
        k = SYMBOLS.index(ch)
        n = int(sr * rng.uniform(0.14, 0.18))
        t = np.arange(n) / sr
        tone = np.sin(2 * np.pi * LOW[k % 5] * t + rng.uniform(0, 6.28)) + np.sin(2 * np.pi * HIGH[k // 5] * t + rng.uniform(0, 6.28))
        env = np.minimum(1.0, np.minimum(np.arange(n), n - 1 - np.arange(n)) / (0.01 * sr))
        parts.append(tone * env * rng.uniform(0.6, 1.0))
        parts.append(np.zeros(int(sr * rng.uniform(0.03, 0.05))))
    x = np.concatenate(parts) * 6000.0
    x += rng.randn(x.shape[0]) * 60.0
    return np.clip(x, -32768, 32767).astype("<i2")


def write_manifest(dirname, name, rng, count):
    rows = ["id,duration,wav,transcript"]
    for i in range(count):
        text = "".join(rng.choice(list(SYMBOLS), int(rng.randint(3, 11))))
        pcm = synth(rng, text)
        p = os.path.join(dirname, "%s%04d.wav" % (name, i))
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(pcm.tobytes())
        rows.append("%d,%.3f,%s,%s" % (i, pcm.shape[0] / 16000.0, p, text))
    path = os.path.join(dirname, name + ".csv")
    with open(path, "w") as fh:
        fh.write("\n".join(rows) + "\n")
    return path


def config(dirname, train_csv, test_csv, epochs, blocks, batch, lr, warmup, ctc_weight=0.3, d_model=256, speed_perturb=False):
    enc = dict(output_size=d_model, attention_heads=d_model // 64, linear_units=2048, num_blocks=blocks, dropout_rate=0.1, positional_dropout_rate=0.1,
               attention_dropout_rate=0, input_layer="conv2d", normalize_before=True, cnn_module_kernel=15, activation_type="swish",
               pos_enc_layer_type="rel_pos", feature_norm=True)
    dec = dict(attention_heads=d_model // 64, linear_units=2048, num_blocks=1, dropout_rate=0.1, positional_dropout_rate=0.1,
               self_attention_dropout_rate=0, src_attention_dropout_rate=0)
    collate = dict(feature_extraction_conf=dict(feature_type="fbank", mel_bins=80, frame_shift=10, frame_length=25, using_pitch=False),
                   feature_dither=0.0, use_speed_perturb=bool(speed_perturb), use_spec_aug=True,
                   spec_aug_conf=dict(warp_for_time=False, num_t_mask=1, num_f_mask=1, prop_mask_t=0.1, prop_mask_f=0.1, max_t=8, max_f=6,
                                      max_w=80),
                   use_dynamic_chunk=False, use_dynamic_left_chunk=False, decoding_chunk_size=0, static_chunk_size=0,
                   num_decoding_left_chunks=-1)
    limits = "144, 204, 288, 400"
    ds = dict(max_length=400, min_length=0, token_max_length=30, token_min_length=1, batch_type="bucket", frame_bucket_limit=limits,
              batch_bucket_limit=", ".join([str(batch)] * 4), batch_factor=1, shuffle=True)
    with open(os.path.join(dirname, "lang_char.txt"), "w") as fh:
        # ids = line index + 2 is what predict.py:146-154's `w += 2` needs to print the right character; the same lines make the token
        # whose index + 2 equals eos = len(char_dict) - 1 end the hypothesis (in the shipped AISHELL dictionary that is one rare
        # character): two unused symbols take that place here
        fh.write("".join("%s %d\n" % (s, i + 2) for i, s in enumerate(["<blank>", "<unk>"] + list(SYMBOLS) + ["<x>", "<y>", "<sos/eos>"])))
    return dict(encoder="conformer", encoder_conf=enc, decoder="transformer", decoder_conf=dec,
                model_conf=dict(ctc_weight=ctc_weight, lsm_weight=0.1, length_normalized_loss=False), collate_conf=collate, dataset_conf=ds,
                test_dataset_conf=dict(ds, shuffle=False), grad_clip=5, accum_grad=1, max_epoch=epochs, log_interval=100, optim="adam",
                optim_conf=dict(lr=lr), scheduler="warmuplr", scheduler_conf=dict(warmup_steps=warmup), cmvn_file="", is_json_cmvn=True,
                exp_name=os.path.join(dirname, "exp"), train_data=train_csv, eval_data=test_csv, test_data=test_csv,
                dict=os.path.join(dirname, "lang_char.txt"), save_checkpoint=True, save_checkpoint_epochs=epochs, save_checkpoint_steps=460,
                keep_checkpoint_max=30, save_checkpoint_path="./", device_target="Ascend", is_distributed=False, mixed_precision=True,
                resume_ckpt="", save_graphs=False, training_with_eval=False, decode_mode="ctc_greedy_search")


def batched_decode_mismatches(cfg, batch_sizes=(7, 16, 48)):
    """Greedy CTC hypotheses of the held-out files decoded in padded batches (lengths and masks as a loader would pass them) that differ
    from the one-utterance-at-a-time decode of predict.py: padding and masks must not move a trained model's output."""
    import torch

    from mindaudio_amd.conformer import predict as P
    from mindaudio_amd.conformer.asr_model import CTCGreedySearch, ctc_greedy_search
    from mindaudio_amd.conformer.dataset import compute_fbank_feats_batch
    from mindaudio_amd.conformer.train import build_model
    from mindaudio_amd.data.io import read
    from mindaudio_amd.utils.ckpt import load_mindspore_checkpoint

    dev = torch.device("cuda", torch.cuda.current_device())
    _, _, vocab, _ = P.load_language_dict(cfg["dict"])
    model = build_model(cfg, 80, vocab, dev)
    load_mindspore_checkpoint(model, os.path.join(cfg["exp_name"], "model", cfg["decode_ckpt"]), strict=False)
    net = CTCGreedySearch(model.eval())
    samples = P.predict_samples(cfg["test_data"], cfg["dict"], cfg["dataset_conf"])
    waves = [np.asarray(read(p)[0], np.float32) * (1 << 15) for _, p, _, _ in samples]

    def decode(ws):
        host = np.zeros((len(ws), max(w.shape[0] for w in ws)), np.float32)
        for i, w in enumerate(ws):
            host[i, :w.shape[0]] = w
        f, nfr = compute_fbank_feats_batch(host, [w.shape[0] for w in ws], sample_rate=16000, frame_len=25, frame_shift=10, mel_bin=80)
        tmax = int(nfr.max())
        m = torch.zeros(len(ws), 1, tmax, device=dev)
        for i, n in enumerate(nfr.tolist()):
            m[i, 0, :n] = 1
        return ctc_greedy_search(net, f[:, :tmax].contiguous(), m, None)[0]

    single = [decode([w])[0] for w in waves]
    bad = 0
    for bs in batch_sizes:
        for lo in range(0, len(waves), bs):
            bad += sum(h != s for h, s in zip(decode(waves[lo:lo + bs]), single[lo:lo + bs]))
    return bad


def run(epochs=40, blocks=2, train_utts=256, test_utts=24, batch=32, lr=1e-3, warmup=60, seed=0, dirname=None, log=None, ctc_weight=0.3,
        d_model=256, speed_perturb=False, with_eval=False, cmvn=False, fp32=False, dynamic_chunk=False, resume_at=0, batch_check=False):
    """cmvn: global_cmvn.json from conformer.compute_cmvn_stats over the training files (GlobalCMVN in the model, train and predict);
    fp32: mixed_precision False (the float32 validation engine); dynamic_chunk: use_dynamic_chunk (a (B, T, T) chunk mask per batch);
    resume_at = E: stop after E epochs, resume from that checkpoint for the rest (the resumed run's first losses are reported)."""
    from mindaudio_amd.conformer import predict as P
    from mindaudio_amd.conformer import train as T

    import random

    random.seed(seed)  # (SpecAugment's / speed perturbation's draws: the reference leaves Python's generator unseeded; this tool's
    #                     runs are meant to repeat)
    dirname = dirname or tempfile.mkdtemp(prefix="ma_recipe_")
    os.makedirs(dirname, exist_ok=True)
    rng = np.random.RandomState(seed)
    train_csv = write_manifest(dirname, "train", rng, train_utts)
    test_csv = write_manifest(dirname, "test", rng, test_utts)
    cfg = config(dirname, train_csv, test_csv, epochs, blocks, batch, lr, warmup, ctc_weight, d_model, speed_perturb)
    tlog = []
    if cmvn:
        from mindaudio_amd.conformer import compute_cmvn_stats as CM

        cfg["cmvn_file"] = os.path.join(dirname, "global_cmvn.json")
        CM.main(train_csv, cfg["cmvn_file"])
    if fp32:
        cfg["mixed_precision"] = False
    if dynamic_chunk:
        cfg["collate_conf"]["use_dynamic_chunk"] = True
    resumed_first = None
    if resume_at:
        cfg1 = dict(cfg, max_epoch=resume_at, save_checkpoint_epochs=resume_at)
        recs1 = T.train(cfg1, log=lambda _l: None)
        ck = sorted(n for n in os.listdir(os.path.join(cfg["exp_name"], "model")) if n.startswith("CKP-%d_" % resume_at) and n.endswith(".ckpt"))[-1]
        cfg["resume_ckpt"] = os.path.join(cfg["exp_name"], "model", ck)
    if with_eval:  # train.py's EvalCallback: evaluation loss per epoch, conformer_<e>_<s>.ckpt, the average of the 30 best at the end
        cfg.update(training_with_eval=True, save_checkpoint_epochs=1)
    t0 = time.perf_counter()
    recs = T.train(cfg, log=log or tlog.append)
    t_train = time.perf_counter() - t0
    steps = len(recs) // max(1, epochs - resume_at)
    if resume_at:
        ep = lambda rs, e: float(np.mean([r["loss"] for r in rs if r["epoch"] == e]))  # noqa: E731  (batches differ in size: epoch means)
        resumed_first = dict(first_epoch_mean=round(ep(recs1, 1), 2), before=round(ep(recs1, resume_at), 2), after=round(ep(recs, resume_at + 1), 2),
                             first_epoch=recs[0]["epoch"], lr_before=recs1[-1]["lr"], lr_after=recs[0]["lr"])
    if with_eval:
        cfg["decode_ckpt"] = "conformer_avg_30.ckpt"
    else:
        ckpts = sorted(n for n in os.listdir(os.path.join(cfg["exp_name"], "model")) if n.startswith("CKP-%d_" % epochs) and n.endswith(".ckpt"))
        cfg["decode_ckpt"] = ckpts[-1]
    plog = []
    try:
        cer, results = P.predict(cfg, log=plog.append)
    except ValueError as e:  # predict.py:164-165: an empty hypothesis stops the reference's script too
        cer, results = float("nan"), [str(e)] + plog[-3:]
    losses = [r["loss"] for r in recs]
    mism = batched_decode_mismatches(cfg) if batch_check and isinstance(results, list) and cer == cer else None
    import torch

    peak_gb = round(torch.cuda.max_memory_allocated() / 2**30, 2) if torch.cuda.is_available() else None
    return dict(steps=len(recs), steps_per_epoch=steps, seconds_training=round(t_train, 1), first_losses=[round(v, 2) for v in losses[:3]],
                last_losses=[round(v, 2) for v in losses[-3:]], overflow_steps=int(sum(bool(r.get("overflow")) for r in recs)),
                peak_gb=peak_gb, eval_losses=[float(ln.split("Average Eval Loss: ")[1].split(",")[0]) for ln in tlog
                                              if ln.startswith("[EvalCallback] Epoch ")][::max(1, epochs // 6)] if with_eval else None,
                decode_ckpt=cfg["decode_ckpt"], resumed=resumed_first, batched_decode_mismatches=mism, held_out_cer=cer, held_out=results[:4] if isinstance(results, list) else results)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--blocks", type=int, default=2)
    ap.add_argument("--utts", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--ctc-weight", type=float, default=0.3)
    ap.add_argument("--d-model", type=int, default=256)
    ap.add_argument("--speed-perturb", action="store_true")
    ap.add_argument("--with-eval", action="store_true", help="training_with_eval: decode from the averaged checkpoint")
    ap.add_argument("--cmvn", action="store_true")
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--dynamic-chunk", action="store_true")
    ap.add_argument("--resume-at", type=int, default=0)
    a = ap.parse_args()
    print(json.dumps(run(a.epochs, a.blocks, a.utts, 24, a.batch, a.lr, a.warmup, ctc_weight=a.ctc_weight, d_model=a.d_model,
                         speed_perturb=a.speed_perturb, with_eval=a.with_eval, cmvn=a.cmvn, fp32=a.fp32, dynamic_chunk=a.dynamic_chunk,
                         resume_at=a.resume_at)))


if __name__ == "__main__":
    main()
