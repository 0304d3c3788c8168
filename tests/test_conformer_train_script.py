"""mindaudio_amd.conformer.train — the counterpart of examples/conformer/train.py on the reference's yaml schema.
CPU: the reference's own conformer.yaml keys are all read (a copy of its schema with small sizes is written here: the reference file
itself is not available on the GPU box), the loop walks a two-batch "dataset", passes the 11 columns in the reference's order to the
step, and prints TimeMonitor's line (mindaudio/utils/callback.py:67-97).  GPU: two real optimizer steps on a two-utterance manifest."""
import os
import re
import wave

import numpy as np
import pytest
import torch
import yaml

from mindaudio_amd.conformer import train as T

HERE = os.path.dirname(os.path.abspath(__file__))

# the schema of examples/conformer/conformer.yaml (every key of the shipped file), small sizes
YAML = """
encoder: conformer
encoder_conf:
    output_size: 256
    attention_heads: 4
    linear_units: 256
    num_blocks: 1
    dropout_rate: 0.1
    positional_dropout_rate: 0.1
    attention_dropout_rate: 0
    input_layer: conv2d
    normalize_before: True
    cnn_module_kernel: 15
    activation_type: 'swish'
    pos_enc_layer_type: 'rel_pos'
    feature_norm : True
decoder: transformer
decoder_conf:
    attention_heads: 4
    linear_units: 256
    num_blocks: 1
    dropout_rate: 0.1
    positional_dropout_rate: 0.1
    self_attention_dropout_rate: 0
    src_attention_dropout_rate: 00
model_conf:
    ctc_weight: 0.3
    lsm_weight: 0.1
    length_normalized_loss: False
collate_conf:
    feature_extraction_conf:
        feature_type: 'fbank'
        mel_bins: 80
        frame_shift: 10
        frame_length: 25
        using_pitch: False
    feature_dither: 0.0
    use_speed_perturb: False
    use_spec_aug: True
    spec_aug_conf:
        warp_for_time: False
        num_t_mask: 2
        num_f_mask: 2
        prop_mask_t: 0.1
        prop_mask_f: 0.1
        max_t: 50
        max_f: 10
        max_w: 80
    use_dynamic_chunk: False
    use_dynamic_left_chunk: False
    decoding_chunk_size: 0
    static_chunk_size: 0
    num_decoding_left_chunks: -1
dataset_conf:
    max_length: 3000
    min_length: 0
    token_max_length: 30
    token_min_length: 1
    batch_type: 'bucket'
    frame_bucket_limit: '144, 204, 288, 400, 512, 600, 712, 800, 912, 1024'
    batch_bucket_limit: '2, 2, 2, 2, 2, 2, 2, 2, 2, 2'
    batch_factor: 1
    shuffle: True
grad_clip: 5
accum_grad: 1
max_epoch: 2
log_interval: 100
optim: adam
optim_conf:
    lr: 0.001
scheduler: warmuplr
scheduler_conf:
    warmup_steps: 25000
cmvn_file: ""
is_json_cmvn: True
exp_name: default
train_data: "train.csv"
eval_data: "dev.csv"
save_checkpoint: False
save_checkpoint_epochs: 1
save_checkpoint_steps: 460
keep_checkpoint_max: 30
save_checkpoint_path: "./"
device_target: "Ascend"
is_distributed: False
mixed_precision: True
resume_ckpt: ""
save_graphs: False
training_with_eval: False
"""

LINE = re.compile(r"^\[Train\] Epoch: \[(\d+)/(\d+)\], Step: \[(\d+)/(\d+)\], Step Time: \d+\.\d{4} sec, lr: \d+\.\d{6}, "
                  r"Total Loss: -?\d+\.\d{4}, (Overflow: True, )?Scale: \d+, Rank: (\d+)\.$")


def _cfg(tmp_path, **over):
    p = tmp_path / "conformer.yaml"
    p.write_text(YAML)
    return T.load_config(str(p), over)


def test_format_is_time_monitors_line():
    a = T.format_step_line(1, 240, 0, 460, 0.0085, 4e-8, 123.45678, 1024.0, 0)
    assert a == "[Train] Epoch: [1/240], Step: [1/460], Step Time: 0.0085 sec, lr: 0.000000, Total Loss: 123.4568, Scale: 1024, Rank: 0."
    b = T.format_step_line(3, 240, 461, 460, 1.25, 1e-3, 7.0, 512.0, 5, overflow=True)
    assert b == "[Train] Epoch: [3/240], Step: [2/460], Step Time: 1.2500 sec, lr: 0.001000, Total Loss: 7.0000, Overflow: True, Scale: 512, Rank: 5."
    assert LINE.match(a) and LINE.match(b)


def test_loop_on_the_reference_schema_with_stubs(tmp_path):
    cfg = _cfg(tmp_path, train_data="x.csv", dict="lang_char.txt", exp_name=str(tmp_path / "exp"), save_checkpoint=True)
    seen = {}

    class FakeData:
        def get_dataset_size(self):
            return 2

        def __iter__(self):
            for k in range(2):
                yield tuple(torch.full((2, 3), float(10 * k + i)) for i in range(11))

    def dataset_factory(data_file, dict_file, collate_conf, dataset_conf, rank, group_size, number_workers):
        seen.update(data_file=data_file, dict_file=dict_file, collate_conf=collate_conf, dataset_conf=dataset_conf, rank=rank, group=group_size)
        return 11, FakeData()

    class FakeStep:
        def __init__(self):
            self.calls = []

        def step(self, *cols):
            assert len(cols) == len(T.COLUMNS) == 11
            self.calls.append([float(c[0, 0]) for c in cols])
            n = len(self.calls)
            return torch.tensor(100.0 / n), n == 2, torch.tensor(1024.0 if n < 3 else 512.0), n == 2, torch.tensor(1e-3 * n)

    fake = FakeStep()
    made = {}

    def model_factory(config, input_dim, vocab, device):
        made.update(input_dim=input_dim, vocab=vocab, conf=config["encoder_conf"])
        return torch.nn.Linear(2, 2)

    lines = []
    recs = T.train(cfg, rank=1, world=2, device=torch.device("cpu"), log=lines.append, dataset_factory=dataset_factory,
                   model_factory=model_factory, step_factory=lambda m, c, r, w, pg: fake)
    # the data-set call sees the reference's keys (the bucket limits as the yaml's comma-separated strings, parsed by the data set)
    assert seen["data_file"] == "x.csv" and seen["dict_file"] == "lang_char.txt" and (seen["rank"], seen["group"]) == (1, 2)
    assert seen["dataset_conf"]["frame_bucket_limit"].startswith("144, 204, 288") and seen["dataset_conf"]["batch_bucket_limit"].startswith("2,")
    assert seen["collate_conf"]["spec_aug_conf"]["max_t"] == 50 and seen["collate_conf"]["feature_extraction_conf"]["mel_bins"] == 80
    assert made["input_dim"] == 80 and made["vocab"] == 11 and made["conf"]["cnn_module_kernel"] == 15
    # 2 epochs x 2 batches, columns in the reference's order (column i of batch k carries 10 k + i)
    assert len(recs) == 4 and fake.calls[1] == [10.0 + i for i in range(11)]
    step_lines = [ln for ln in lines if ln.startswith("[Train]")]
    assert len(step_lines) == 4 and all(LINE.match(ln) for ln in step_lines)
    got = [LINE.match(ln).groups() for ln in step_lines]
    assert [(g[0], g[2]) for g in got] == [("1", "1"), ("1", "2"), ("2", "1"), ("2", "2")] and all(g[1] == "2" and g[3] == "2" for g in got)
    assert got[1][4] == "Overflow: True, " and got[0][4] is None and all(g[5] == "1" for g in got)
    assert "Training dataset has 2 steps in each epoch." in lines and "Training start." in lines
    assert recs[1]["overflow"] is True and recs[2]["scale"] == 512.0 and abs(recs[3]["lr"] - 4e-3) < 1e-9


def test_unsupported_keys_fail_loudly(tmp_path):
    cfg = _cfg(tmp_path, scheduler="cosine")
    with pytest.raises(ValueError, match="Only 'none', and 'warmuplr' are supported"):  # examples/conformer/train.py:135
        T.build_step(torch.nn.Linear(2, 2), cfg, 0, 1)


class _EvalData:
    """Three batches of 2, 3 and 1 "utterances" (first column), like a BucketASRDataset iterable."""

    def __init__(self):
        self.batches = [(torch.full((n, 4), float(v)),) for n, v in ((2, 1.0), (3, 2.0), (1, 4.0))]

    def get_dataset_size(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


class _LossNet(torch.nn.Module):
    """forward(x) = mean(x) * w: a "loss" the evaluation can be checked against by hand."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([1.0]))
        self.register_buffer("running_mean", torch.zeros(3))

    def forward(self, x):
        return (x.mean() * self.w).reshape(())


def test_eval_callback_logs_saves_and_averages_like_the_reference(tmp_path):
    """mindaudio/utils/callback.py:256-447: conformer_init.ckpt; every run_interval steps the utterance-weighted mean loss, a log line,
    conformer_<epoch>_<step>.ckpt + .yaml; at the end conformer_avg_<n>.ckpt = the element-wise mean of the n lowest-loss files."""
    import yaml

    from mindaudio_amd.utils.ckpt import read_mindspore_ckpt

    net, lines = _LossNet(), []

    class _Eng:
        synced = 0

        def sync_to_module(self):
            _Eng.synced += 1

    cb = T.EvalCallback(net, _Eng(), _EvalData(), torch.device("cpu"), 1, 0, 2, str(tmp_path / "model"), lines.append, num_best_ckpt=2)
    cb.begin()
    assert os.path.exists(str(tmp_path / "model" / "conformer_init.ckpt"))
    weights = {2: 1.0, 4: 3.0, 6: 2.0}
    for step in range(1, 7):
        if step in weights:
            with torch.no_grad():
                net.w.fill_(weights[step])
        cb.step_end((step + 1) // 2, 3, step)
    cb.end()
    base = (2 * 1.0 + 3 * 2.0 + 1 * 4.0) / 6  # utterance-weighted mean of the batch losses at w = 1
    evals = [ln for ln in lines if "Average Eval Loss" in ln]
    assert [ln.split("Average Eval Loss: ")[1].split(",")[0] for ln in evals] == ["%.4f" % (base * w) for w in (1.0, 3.0, 2.0)]
    assert evals[0].startswith("[EvalCallback] Epoch 1/3,") and evals[2].startswith("[EvalCallback] Epoch 3/3,")
    assert net.training and _Eng.synced >= 4  # evaluation mode is left again; the engine's weights are synchronised before every read
    for e, st, w in ((1, 2, 1.0), (2, 4, 3.0), (3, 6, 2.0)):
        ck = read_mindspore_ckpt(str(tmp_path / "model" / ("conformer_%d_%d.ckpt" % (e, st))))
        assert float(ck["w"][0]) == w and int(ck["epoch_num"]) == e and "running_mean" in ck
        info = yaml.safe_load(open(str(tmp_path / "model" / ("conformer_%d_%d.yaml" % (e, st)))))
        assert abs(info["loss"] - base * w) < 1e-6 and info["time"] >= 0
    avg = read_mindspore_ckpt(str(tmp_path / "model" / "conformer_avg_2.ckpt"))
    assert abs(float(avg["w"][0]) - 1.5) < 1e-7  # the two lowest losses: w = 1 and w = 2
    assert lines[-1].startswith("[EvalCallback] [After training] Total Eval Time: 0h 0m ")


class _HostStep:
    """The host side of ConformerCTCTrainStep (schedule index, start_steps, sync_to_module) without device work: what the resume
    logic of the loop talks to."""

    def __init__(self, model, config, start_steps=0):
        from mindaudio_amd.train.engine import asr_warmup_lr

        self.model, self.start_steps, self.global_step, self._lr = model, start_steps, 0, asr_warmup_lr
        self.sched = config.get("scheduler", "warmuplr")
        self.base, self.warm = float(config["optim_conf"]["lr"]), int(config["scheduler_conf"]["warmup_steps"])

    def step(self, *cols):
        lr = self.base if self.sched == "none" else self._lr(self.global_step, self.base, self.warm, self.start_steps)
        self.global_step += 1
        with torch.no_grad():  # "training": every parameter and the BatchNorm statistics move
            for p in self.model.parameters():
                p.add_(0.01)
            bn = self.model.encoder.encoders[0].conv_module.norm
            bn.running_mean.add_(0.25)
            bn.running_var.mul_(1.5)
        return torch.tensor(50.0), False, torch.tensor(1024.0), False, torch.tensor(lr, dtype=torch.float64)

    def sync_to_module(self):
        pass


def test_resume_continues_the_schedule_and_checkpoints_carry_reference_names(tmp_path):
    """examples/conformer/train.py:117-133,172-179: epoch_num from the checkpoint -> ASRWarmupLR(start_steps = epoch_num * steps_size),
    max_epoch - epoch_num epochs; the file holds the reference's parameter names and layouts incl. the BatchNorm moving statistics and
    loads strictly."""
    from mindaudio_amd.utils import ckpt as CK

    class Data:
        def get_dataset_size(self):
            return 3

        def __iter__(self):
            for k in range(3):
                yield tuple(torch.zeros(1, 1) for _ in range(11))

    def run(**over):
        cfg = _cfg(tmp_path, train_data="x.csv", dict="d.txt", exp_name=str(tmp_path / "exp"), save_checkpoint=True, **over)
        torch.manual_seed(5)
        lines = []
        holder = {}

        def step_factory(m, c, r, w, pg, start_steps=0):
            holder["model"], holder["start_steps"] = m, start_steps
            return _HostStep(m, c, start_steps)

        recs = T.train(cfg, device=torch.device("cpu"), log=lines.append, dataset_factory=lambda *a, **k: (20, Data()),
                       step_factory=step_factory)
        return recs, lines, holder

    full, _, _ = run(max_epoch=3)
    assert [r["epoch"] for r in full] == [1, 1, 1, 2, 2, 2, 3, 3, 3] and full[0]["lr"] == 0.0
    first, _, h1 = run(max_epoch=1)
    path = str(tmp_path / "exp" / "model" / "CKP-1_3.ckpt")
    raw = CK.read_mindspore_ckpt(path)
    assert int(raw["epoch_num"]) == 1 and CK.read_epoch_num(path) == 1
    # the reference's names and layouts
    assert raw["encoder.encoders.0.conv_module.depthwise_conv.conv1d.weight"].shape == (256, 1, 1, 15)
    assert "encoder.encoders.0.conv_module.norm.moving_variance" in raw and "encoder.encoders.0.conv_module.norm.gamma" in raw
    assert "encoder.embed.conv.2.conv2d.weight" in raw and "encoder.encoders.0.feed_forward.w_1.dense.weight" in raw
    assert "decoder.embed.0.embedding_table" in raw and "ctc.ctc_lo.weight" in raw
    # a strict load into a fresh model gives the trained model back, BatchNorm statistics included
    fresh = T.build_model(_cfg(tmp_path), 80, 20, torch.device("cpu"))
    missing, unexpected = CK.load_mindspore_checkpoint(fresh, path, strict=True)
    assert not missing and not unexpected
    a, b = h1["model"].state_dict(), fresh.state_dict()
    for k in a:
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(a[k], b[k]), k
    assert float(b["encoder.encoders.0.conv_module.norm.running_mean"].mean()) == pytest.approx(0.75)
    # resumed: the schedule continues, two epochs remain, epoch labels and checkpoint names go on from 2
    rest, lines, h2 = run(max_epoch=3, resume_ckpt=path)
    assert h2["start_steps"] == 3 and "Successfully loading the pre-trained model" in lines
    assert [r["epoch"] for r in rest] == [2, 2, 2, 3, 3, 3]
    assert [r["lr"] for r in rest] == [r["lr"] for r in full[3:]] and rest[0]["lr"] > 0.0
    assert os.path.exists(str(tmp_path / "exp" / "model" / "CKP-3_3.ckpt"))
    assert CK.read_epoch_num(str(tmp_path / "exp" / "model" / "CKP-3_3.ckpt")) == 3
    sl = [ln for ln in lines if ln.startswith("[Train]")]
    assert sl[0].startswith("[Train] Epoch: [2/3], Step: [1/3]")
    # scheduler: none = Adam at the constant lr (train.py:126-127)
    const, _, _ = run(max_epoch=1, scheduler="none")
    assert [r["lr"] for r in const] == pytest.approx([1e-3] * 3)


@pytest.mark.gpu
def test_two_real_steps_on_a_two_utterance_manifest(tmp_path):
    """create_dataset -> create_asr_model (hybrid CTC / attention, the shipped yaml's model_conf) -> ConformerCTCTrainStep: two optimizer
    steps from the command line entry point; finite losses, the warm-up schedule's learning rates, one line per step."""
    src = os.path.join(HERE, "golden", "BAC009S0002W0122.wav")
    with wave.open(src, "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    paths = []
    for i, n in enumerate((40000, 44000)):  # 248 and 273 frames: one bucket (<= 288), one batch of two
        p = str(tmp_path / ("utt%d.wav" % i))
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(pcm[:n].tobytes())
        paths.append(p)
    (tmp_path / "lang_char.txt").write_text("".join("%s %d\n" % (ch, i) for i, ch in enumerate(["<blank>", "<unk>", "a", "b", "c", "d", "<sos/eos>"])))
    (tmp_path / "train.csv").write_text("id,duration,wav,transcript\n0,2.5,%s,abca\n1,2.75,%s,dcb\n" % tuple(paths))
    cfg = _cfg(tmp_path, train_data=str(tmp_path / "train.csv"), dict=str(tmp_path / "lang_char.txt"), max_epoch=2,
               exp_name=str(tmp_path / "exp"), save_checkpoint=True)
    lines = []
    recs = T.train(cfg, log=lines.append)
    assert len(recs) == 2 and all(np.isfinite(r["loss"]) and r["loss"] > 0 for r in recs)
    from mindaudio_amd.train.engine import asr_warmup_lr

    assert abs(recs[0]["lr"] - asr_warmup_lr(0, 1e-3, 25000)) < 1e-12 and abs(recs[1]["lr"] - asr_warmup_lr(1, 1e-3, 25000)) < 1e-12
    assert sum(ln.startswith("[Train]") and bool(LINE.match(ln)) for ln in lines) == 2
    # resumed from the first epoch's checkpoint with the real engine: one epoch remains, the schedule continues at step 1, and the
    # step sees the trained weights and BatchNorm statistics (same batch, dropout stream of a fresh engine: the loss is near epoch
    # 2's, far from epoch 1's only if training moved it - so only finiteness and the schedule are asserted)
    ck = os.path.join(str(tmp_path / "exp"), "model", "CKP-1_1.ckpt")
    assert os.path.exists(ck)
    cfg2 = dict(cfg, resume_ckpt=ck, save_checkpoint=False)
    lines2 = []
    rest = T.train(cfg2, log=lines2.append)
    assert len(rest) == 1 and rest[0]["epoch"] == 2 and np.isfinite(rest[0]["loss"])
    assert abs(rest[0]["lr"] - asr_warmup_lr(1, 1e-3, 25000)) < 1e-12 and rest[0]["lr"] > 0
    assert any(ln.startswith("[Train] Epoch: [2/2], Step: [1/1]") for ln in lines2)


@pytest.mark.gpu
def test_three_epochs_on_a_ragged_manifest_with_long_transcripts(tmp_path):
    """What the two-utterance test cannot see: 26 utterances of 0.8 ... 5.6 s in five frame buckets, transcripts of 2 ... 44 tokens
    (longer than one 32-query tile of the decoder's attention kernels), speed perturbation and SpecAugment on, the hybrid loss - so
    every batch has another shape AND another label width: launch tables recorded, replayed and mixed with walked decoders, plans
    per shape, checkpoints, resume.  Three epochs from the command-line entry point: every loss finite, the last epoch's mean loss
    below the first's, one checkpoint per epoch, and a run resumed from the second reproduces the number of remaining steps."""
    src = os.path.join(HERE, "golden", "BAC009S0002W0122.wav")
    with wave.open(src, "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    rng = np.random.RandomState(7)
    chars = [chr(ord("a") + i) for i in range(20)]
    (tmp_path / "lang_char.txt").write_text("".join("%s %d\n" % (ch, i) for i, ch in enumerate(["<blank>", "<unk>"] + chars + ["<sos/eos>"])))
    rows = ["id,duration,wav,transcript"]
    for i in range(26):
        n = int(rng.randint(12800, 90000))
        p = str(tmp_path / ("utt%02d.wav" % i))
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(np.resize(pcm, n).tobytes())
        ntok = int(rng.randint(2, 45)) if i % 5 else 44  # (several transcripts beyond 31 tokens)
        ntok = min(ntok, n // 160 // 4 - 3)              # (CTC: at most T' labels)
        rows.append("%d,%.2f,%s,%s" % (i, n / 16000.0, p, "".join(rng.choice(chars, ntok))))
    (tmp_path / "train.csv").write_text("\n".join(rows) + "\n")
    over = dict(train_data=str(tmp_path / "train.csv"), dict=str(tmp_path / "lang_char.txt"), max_epoch=3, exp_name=str(tmp_path / "exp"),
                save_checkpoint=True)
    cfg = _cfg(tmp_path, **over)
    cfg["dataset_conf"].update(token_max_length=60, batch_bucket_limit="4, 4, 4, 4, 4, 4, 4, 4, 4, 4")
    cfg["collate_conf"].update(use_speed_perturb=True)
    cfg["scheduler_conf"]["warmup_steps"] = 10
    cfg["optim_conf"]["lr"] = 2e-3
    lines = []
    recs = T.train(cfg, log=lines.append)
    assert recs and all(np.isfinite(r["loss"]) and r["loss"] > 0 for r in recs)
    per_epoch = {e: [r["loss"] for r in recs if r["epoch"] == e] for e in (1, 2, 3)}
    steps = len(per_epoch[1])
    assert steps >= 7 and len(per_epoch[2]) == steps and len(per_epoch[3]) == steps
    assert np.mean(per_epoch[3]) < np.mean(per_epoch[1]), per_epoch
    ck = [os.path.join(str(tmp_path / "exp"), "model", "CKP-%d_%d.ckpt" % (e, steps)) for e in (1, 2, 3)]
    assert all(os.path.exists(c) for c in ck)
    rest = T.train(dict(cfg, resume_ckpt=ck[1], save_checkpoint=False), log=lambda _l: None)
    assert len(rest) == steps and all(r["epoch"] == 3 and np.isfinite(r["loss"]) for r in rest)


@pytest.mark.gpu
def test_training_with_eval_then_predict_from_the_averaged_checkpoint(tmp_path):
    """The reference's whole recipe on one small manifest: train.py with training_with_eval (EvalCallback: evaluation loss per epoch,
    conformer_<e>_<s>.ckpt, conformer_avg_30.ckpt at the end) and predict.py (ctc_greedy_search) from the averaged checkpoint."""
    from mindaudio_amd.conformer import predict as P
    from mindaudio_amd.utils.ckpt import read_mindspore_ckpt

    src = os.path.join(HERE, "golden", "BAC009S0002W0122.wav")
    with wave.open(src, "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    rng = np.random.RandomState(11)
    chars = [chr(ord("a") + i) for i in range(12)]
    (tmp_path / "lang_char.txt").write_text("".join("%s %d\n" % (ch, i + 2) for i, ch in enumerate(["<blank>", "<unk>"] + chars + ["<sos/eos>"])))
    for name, count in (("train", 12), ("dev", 5)):
        rows = ["id,duration,wav,transcript"]
        for i in range(count):
            n = int(rng.randint(16000, 60000))
            p = str(tmp_path / ("%s%02d.wav" % (name, i)))
            with wave.open(p, "wb") as w:
                w.setnchannels(1)
                w.setsampwidth(2)
                w.setframerate(16000)
                w.writeframes(np.resize(pcm, n).tobytes())
            rows.append("%d,%.2f,%s,%s" % (i, n / 16000.0, p, "".join(rng.choice(chars, int(rng.randint(2, 9))))))
        (tmp_path / (name + ".csv")).write_text("\n".join(rows) + "\n")
    cfg = _cfg(tmp_path, train_data=str(tmp_path / "train.csv"), eval_data=str(tmp_path / "dev.csv"), test_data=str(tmp_path / "dev.csv"),
               dict=str(tmp_path / "lang_char.txt"), max_epoch=3, exp_name=str(tmp_path / "exp"), training_with_eval=True,
               decode_mode="ctc_greedy_search", decode_ckpt="conformer_avg_30.ckpt")
    cfg["dataset_conf"]["batch_bucket_limit"] = "4, 4, 4, 4, 4, 4, 4, 4, 4, 4"
    cfg["test_dataset_conf"] = dict(cfg["dataset_conf"], shuffle=False)
    cfg["scheduler_conf"]["warmup_steps"] = 5
    lines = []
    recs = T.train(cfg, log=lines.append)
    steps = len(recs) // 3
    evals = [ln for ln in lines if ln.startswith("[EvalCallback] Epoch ")]
    assert len(evals) == 3 and all(np.isfinite(float(ln.split("Average Eval Loss: ")[1].split(",")[0])) for ln in evals)
    model_dir = tmp_path / "exp" / "model"
    names = sorted(os.listdir(str(model_dir)))
    assert "conformer_init.ckpt" in names and "conformer_avg_30.ckpt" in names
    assert all(("conformer_%d_%d.ckpt" % (e, e * steps)) in names and ("conformer_%d_%d.yaml" % (e, e * steps)) in names for e in (1, 2, 3))
    assert not any(n.startswith("CKP-") for n in names)  # EvalCallback takes ModelCheckpoint's place (train.py:143-164)
    avg = read_mindspore_ckpt(str(model_dir / "conformer_avg_30.ckpt"))
    one = [read_mindspore_ckpt(str(model_dir / ("conformer_%d_%d.ckpt" % (e, e * steps)))) for e in (1, 2, 3)]
    key = "encoder.encoders.0.feed_forward.w_1.dense.weight"
    assert np.allclose(avg[key], np.mean([c[key] for c in one], axis=0), atol=1e-6)
    # predict.py from the averaged checkpoint: three epochs of a 1-block model decode to garbage or to nothing - the run must get as far
    # as the reference's own complaint about an empty hypothesis, or through
    try:
        mean, results = P.predict(cfg, log=lambda _l: None)
        assert 0 <= mean and len(results) == 5
    except ValueError as e:
        assert "Hypothesis" in str(e)


@pytest.mark.gpu
def test_overlapped_loader_changes_no_loss(tmp_path):
    """conformer/train.py collates batch n + 1 (uploads on a copy stream, device feature kernels behind the step on the compute
    stream) between enqueue_step(n) and finish_step(n).  The same run with a step factory that hides enqueue_step / finish_step - the
    loop then collates and steps one after the other - must log the same losses, step for step, to the last bit."""
    import random

    src = os.path.join(HERE, "golden", "BAC009S0002W0122.wav")
    with wave.open(src, "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    rng = np.random.RandomState(3)
    chars = [chr(ord("a") + i) for i in range(12)]
    (tmp_path / "lang_char.txt").write_text("".join("%s %d\n" % (ch, i) for i, ch in enumerate(["<blank>", "<unk>"] + chars + ["<sos/eos>"])))
    rows = ["id,duration,wav,transcript"]
    for i in range(20):
        n = int(rng.randint(16000, 80000))
        p = str(tmp_path / ("utt%02d.wav" % i))
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(np.resize(pcm, n).tobytes())
        rows.append("%d,%.2f,%s,%s" % (i, n / 16000.0, p, "".join(rng.choice(chars, int(rng.randint(2, 20))))))
    (tmp_path / "train.csv").write_text("\n".join(rows) + "\n")
    cfg = _cfg(tmp_path, train_data=str(tmp_path / "train.csv"), dict=str(tmp_path / "lang_char.txt"), max_epoch=3,
               exp_name=str(tmp_path / "exp"))
    cfg["dataset_conf"]["batch_bucket_limit"] = "4, 4, 4, 4, 4, 4, 4, 4, 4, 4"
    cfg["collate_conf"].update(use_speed_perturb=True)
    cfg["scheduler_conf"]["warmup_steps"] = 5

    class _StepOnly:
        def __init__(self, eng):
            self._eng = eng

        def step(self, *cols):
            return self._eng.step(*cols)

    def serial_factory(model, config, rank, world, process_group=None, start_steps=0):
        return _StepOnly(T.build_step(model, config, rank, world, process_group, start_steps=start_steps))

    runs = []
    for factory in (None, serial_factory):
        random.seed(11)
        np.random.seed(11)
        recs = T.train(cfg, log=lambda _l: None, step_factory=factory)
        runs.append([r["loss"] for r in recs])
    assert len(runs[0]) >= 15 and runs[0] == runs[1]


@pytest.mark.gpu
def test_two_ranks_run_the_script_and_hold_the_same_weights(tmp_path):
    """The training script data-parallel: two processes (gloo on device tensors, one GPU) walk the same shuffled batch order, keep
    batch[rank::2] each, all-reduce the gradients of the hybrid step and must end two epochs with bit-identical weights - with the
    loader overlapped, launch tables recorded, and label widths that differ between the ranks' shards."""
    import socket
    import subprocess
    import sys

    src = os.path.join(HERE, "golden", "BAC009S0002W0122.wav")
    with wave.open(src, "rb") as w:
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    rng = np.random.RandomState(4)
    chars = [chr(ord("a") + i) for i in range(12)]
    (tmp_path / "lang_char.txt").write_text("".join("%s %d\n" % (ch, i) for i, ch in enumerate(["<blank>", "<unk>"] + chars + ["<sos/eos>"])))
    rows = ["id,duration,wav,transcript"]
    for i in range(24):
        n = int(rng.randint(16000, 70000))
        p = str(tmp_path / ("utt%02d.wav" % i))
        with wave.open(p, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(16000)
            w.writeframes(np.resize(pcm, n).tobytes())
        rows.append("%d,%.2f,%s,%s" % (i, n / 16000.0, p, "".join(rng.choice(chars, int(rng.randint(2, 22))))))
    (tmp_path / "train.csv").write_text("\n".join(rows) + "\n")
    (tmp_path / "conformer.yaml").write_text(YAML)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    outs = [str(tmp_path / ("r%d.pt" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "train_worker.py"), str(r), "2", str(port), str(tmp_path), outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for p in procs:
        log, _ = p.communicate(timeout=600)
        assert p.returncode == 0, log[-3000:]
    a, b = (torch.load(o) for o in outs)
    assert len(a["losses"]) == len(b["losses"]) >= 6 and all(np.isfinite(v) for v in a["losses"] + b["losses"])
    assert torch.equal(a["master"], b["master"])
    assert a["losses"] != b["losses"]  # (each rank logs the loss of ITS shard)


@pytest.mark.gpu
def test_the_recipe_learns_a_synthetic_language(tmp_path):
    """End to end, the one property no parity test has: the recipe LEARNS.  tools/recipe_learns.py writes a tone-coded language (every
    symbol a pair of tones, 3 ... 10 symbols per utterance) as wav files, runs conformer.train (device Kaldi fbank, SpecAugment,
    buckets, hybrid CTC / attention step, Adam + warm-up + loss scale, BatchNorm statistics, reference-format checkpoint) for 150 steps
    and conformer.predict (greedy CTC, CER) from the checkpoint on 24 utterances the training never saw: the loss falls by more than
    half and the held-out CER is below 5 % (measured: 0.0 after 120 steps)."""
    import sys

    sys.path.insert(0, os.path.dirname(HERE))
    from tools.recipe_learns import run

    res = run(epochs=15, blocks=2, train_utts=256, test_utts=24, batch=32, dirname=str(tmp_path / "two"))
    assert res["steps"] >= 120 and res["overflow_steps"] == 0, res
    assert np.mean(res["last_losses"]) < 0.5 * np.mean(res["first_losses"]), res
    assert res["held_out_cer"] == res["held_out_cer"] and res["held_out_cer"] <= 0.05, res
    # the same with speed perturbation (the device resampler in the loop), with the pure-CTC model, and on the d_model 512 engine
    # (one launch per reference cell instead of the fused d_model 256 launches)
    for name, kw in (("speed", dict(speed_perturb=True)), ("ctc", dict(ctc_weight=1.0)), ("d512", dict(d_model=512))):
        res = run(epochs=25, blocks=2, train_utts=256, test_utts=24, batch=32, dirname=str(tmp_path / name), **kw)
        assert res["overflow_steps"] == 0 and res["held_out_cer"] == res["held_out_cer"] and res["held_out_cer"] <= 0.05, (name, res)
    # GlobalCMVN from compute_cmvn_stats' json (train and predict), the (B, T, T) chunk masks of use_dynamic_chunk, the float32
    # validation engine (mixed_precision False), and a run stopped after 20 epochs and resumed from its checkpoint (weights, BatchNorm
    # statistics, schedule position: the resumed run's first losses are near the stopped run's last, far below an untrained model's)
    for name, kw in (("cmvn", dict(cmvn=True)), ("chunk", dict(dynamic_chunk=True)), ("fp32", dict(fp32=True)), ("resume", dict(resume_at=20))):
        res = run(epochs=40 if name == "resume" else 25, blocks=2, train_utts=256, test_utts=24, batch=32, dirname=str(tmp_path / name), **kw)
        assert res["overflow_steps"] == 0 and res["held_out_cer"] == res["held_out_cer"] and res["held_out_cer"] <= 0.05, (name, res)
        if name == "resume":
            r = res["resumed"]
            # (epoch means: single batches differ in size and so in loss)
            assert r["first_epoch"] == 21 and 0 < r["lr_after"] < r["lr_before"], res
            assert r["after"] < 1.6 * r["before"] and r["after"] < 0.5 * r["first_epoch_mean"], res
    # training_with_eval: the evaluation loss the EvalCallback logs between training steps follows the training (it stopped at the
    # untrained decoder's until the end of round 6: sync_to_module left the decoder's packed evaluation weights in place), and the
    # averaged checkpoint decodes
    res = run(epochs=40, blocks=2, train_utts=256, test_utts=24, batch=32, dirname=str(tmp_path / "eval"), with_eval=True)
    assert res["decode_ckpt"] == "conformer_avg_30.ckpt" and res["held_out_cer"] <= 0.05, res
    assert res["eval_losses"][-1] < 0.4 * res["eval_losses"][0], res
    # the shipped depth (12 blocks; a deep model sits on the all-blank plateau first: a longer warm-up, 1 000 steps, ~7 s)
    res = run(epochs=100, blocks=12, train_utts=256, test_utts=24, batch=32, lr=5e-4, warmup=300, dirname=str(tmp_path / "twelve"),
              batch_check=True)
    assert res["overflow_steps"] == 0 and np.mean(res["last_losses"]) < 0.1 * np.mean(res["first_losses"]), res
    assert res["held_out_cer"] == res["held_out_cer"] and res["held_out_cer"] <= 0.05, res
    # ... and the trained model's hypotheses do not depend on how the utterances are batched and padded (batches of 7, 16 and all 24
    # against one at a time: masks and padding of the fused evaluation forward on weights that are no longer random)
    assert res["batched_decode_mismatches"] == 0, res
