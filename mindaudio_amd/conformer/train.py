"""ASR training loop on the reference's configuration schema — the counterpart of examples/conformer/train.py:53-179.

    python -m mindaudio_amd.conformer.train --config_path conformer.yaml [--train_data x.csv --dict lang_char.txt ...]

Reads the keys of examples/conformer/conformer.yaml (encoder_conf, decoder_conf, model_conf, collate_conf, dataset_conf, optim_conf,
scheduler / scheduler_conf, cmvn_file / is_json_cmvn, train_data, dict, max_epoch, exp_name, save_checkpoint*, is_distributed,
mixed_precision, resume_ckpt), builds `create_dataset` -> `create_asr_model` -> `ConformerCTCTrainStep` (what the reference
builds as Adam + ASRWarmupLR + DynamicLossScaleUpdateCell(1024, 2, 1000) + TrainOneStepWithLossScaleCell, train.py:126-141) and
prints the fields of the reference's `TimeMonitor` (mindaudio/utils/callback.py:48-99) per step:

    [Train] Epoch: [e/E], Step: [s/S], Step Time: 0.0085 sec, lr: 0.000040, Total Loss: 123.4567, Scale: 1024, Rank: 0.

One process per GPU: under `torchrun` (RANK / WORLD_SIZE / LOCAL_RANK) with is_distributed the gradients are averaged over RCCL
(`mindspore.set_auto_parallel_context(DATA_PARALLEL, gradients_mean=True)`, train.py:70-80) and every rank keeps
batch[rank::world] of the same shuffled batch order (dataset.py:552-553).  Keys the reference's train.py never reads either
(grad_clip, accum_grad, log_interval, optim, device_target, save_graphs, full_graph) are accepted and ignored; `training_with_eval`
runs the reference's EvalCallback (evaluation loss, `conformer_<epoch>_<step>.ckpt`, the averaged checkpoint at the end) in place of
ModelCheckpoint, as train.py:143-164 does; `scheduler` is "warmuplr" or "none" and anything else raises the
reference's ValueError (train.py:126-135).

Resume (train.py:117-133,172-179): `resume_ckpt` is read with the reference's parameter names, `epoch_num` from the file is the
number of finished epochs: the schedule continues at `start_steps = epoch_num * steps_size`, `max_epoch - epoch_num` epochs run, and
the log / checkpoint names count epochs from `epoch_num + 1` (ResumeCallback).  Checkpoints are written under the reference's
parameter names and layouts (`utils.ckpt.to_reference_names`), BatchNorm moving statistics included, with `epoch_num` appended."""
import argparse
import os
import sys
import time

import yaml

COLUMNS = ("xs_pad", "ys_pad", "ys_in_pad", "ys_out_pad", "r_ys_in_pad", "r_ys_out_pad", "xs_masks", "ys_masks", "ys_sub_masks",
           "ys_lengths", "xs_chunk_masks")  # train.py:38-50


def load_config(path, overrides=None):
    """The yaml as a plain dict (mindaudio.utils.config.get_config parses the same file into an attribute dict); `overrides` maps
    top-level keys to replacement values (the reference takes them as --key value command-line options)."""
    with open(path) as fh:
        cfg = yaml.safe_load(fh)
    for k, v in (overrides or {}).items():
        if v is not None:
            cfg[k] = v
    return cfg


def format_step_line(epoch, max_epoch, step, steps_size, seconds, lr, loss, scale, rank, overflow=False):
    """TimeMonitor.step_end's line (callback.py:67-97); `step` counts from 0 over the whole run as TimeMonitor.step does."""
    head = "[Train] Epoch: [%d/%d], Step: [%d/%d], Step Time: %.4f sec, lr: %.6f, Total Loss: %.4f, " % (
        epoch, max_epoch, step % steps_size + 1, steps_size, seconds, lr, loss)
    if overflow:
        return head + "Overflow: %s, Scale: %.0f, Rank: %d." % (str(overflow), scale, rank)
    return head + "Scale: %.0f, Rank: %d." % (scale, rank)


def build_model(config, input_dim, vocab_size, device):
    """creadte_asr_model (asr_model.py:301-352): GlobalCMVN from cmvn_file, ConformerEncoder(**encoder_conf), a TransformerDecoder
    unless ctc_weight == 1, ASRModel(**model_conf)."""
    import torch

    from ..utils.load_files import load_cmvn
    from .asr_model import create_asr_model

    if config.get("encoder", "conformer") != "conformer" or config.get("decoder", "transformer") != "transformer":
        raise NotImplementedError("encoder: conformer / decoder: transformer (asr_model.py:316-337)")
    cmvn = None
    if config.get("cmvn_file"):
        mean, istd = load_cmvn(config["cmvn_file"], config.get("is_json_cmvn", True))
        cmvn = (torch.tensor(mean, dtype=torch.float32), torch.tensor(istd, dtype=torch.float32))
    mc = config.get("model_conf", {})
    model = create_asr_model(input_dim, vocab_size, dict(config.get("encoder_conf") or {}), global_cmvn=cmvn,
                             ctc_weight=float(mc.get("ctc_weight", 0.3)), decoder_conf=dict(config.get("decoder_conf") or {}),
                             lsm_weight=float(mc.get("lsm_weight", 0.0)), length_normalized_loss=bool(mc.get("length_normalized_loss", False)))
    return model.to(device)


def build_step(model, config, rank, world, process_group=None, start_steps=0):
    """Adam(lr = ASRWarmupLR(lr, warmup_steps, start_steps) | the constant lr of `scheduler: none`) +
    DynamicLossScaleUpdateCell(1024, 2, 1000) + TrainOneStepWithLossScaleCell (train.py:111-141) as one ConformerCTCTrainStep; mixed_precision True -> bf16 matmuls with float32 masters (the reference's
    float16 compute_type), False -> the float32 mode."""
    import torch

    from ..train.engine import ConformerCTCTrainStep

    sched = config.get("scheduler", "warmuplr")
    if sched not in ("warmuplr", "none"):
        raise ValueError("Only 'none', and 'warmuplr' are supported.")  # train.py:135
    enc_conf = config.get("encoder_conf") or {}
    return ConformerCTCTrainStep(model, base_lr=float(config["optim_conf"]["lr"]), scheduler=sched, start_steps=int(start_steps),
                                 warmup_steps=int((config.get("scheduler_conf") or {}).get("warmup_steps", 25000)), loss_scale=1024.0,
                                 scale_factor=2.0,
                                 scale_window=1000, dropout_rate=float(enc_conf.get("dropout_rate", 0.1)),
                                 positional_dropout_rate=float(enc_conf.get("positional_dropout_rate", 0.1)), seed=777,
                                 process_group=process_group, world_size=world, rank=rank,
                                 compute_type=None if config.get("mixed_precision", True) else torch.float32)


def train(config, rank=0, world=1, device=None, max_steps=None, log=print, dataset_factory=None, model_factory=None,
          step_factory=None, process_group=None):
    """The loop of train.py:53-179.  Returns the list of per-step records (epoch, step, loss, scale, overflow, lr, seconds).
    The three factories default to the real pieces (create_dataset / build_model / build_step); tests substitute them."""
    import torch

    torch.manual_seed(777)  # set_seed(777), train.py:56
    import numpy as np

    np.random.seed(777)     # (mindspore.set_seed seeds numpy's global generator as well; Python's `random` - SpecAugment's and speed
    #                          perturbation's draws, dataset.py:398-406,493-534 - stays unseeded there and here)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    ds_conf = dict(config["dataset_conf"])  # (frame_bucket_limit / batch_bucket_limit stay the yaml's comma-separated strings: the data
    # set parses them, dataset.py:304-305)
    if dataset_factory is None:
        from .dataset import create_dataset as dataset_factory
    vocab_size, dataset = dataset_factory(config["train_data"], config["dict"], collate_conf=dict(config["collate_conf"]),
                                          dataset_conf=ds_conf, rank=rank, group_size=world, number_workers=1)
    eval_dataset = None
    if config.get("training_with_eval"):
        # train.py:93-106: the evaluation set without speed perturbation and SpecAugment, under test_dataset_conf
        log("Initializing evaluation dataset.")
        eval_collate = dict(config["collate_conf"], use_speed_perturb=False, use_spec_aug=False)
        _, eval_dataset = dataset_factory(config["eval_data"], config["dict"], collate_conf=eval_collate,
                                          dataset_conf=dict(config.get("test_dataset_conf") or ds_conf), rank=rank, group_size=world,
                                          number_workers=1)
    input_dim = int(config["collate_conf"]["feature_extraction_conf"]["mel_bins"])
    steps_size = dataset.get_dataset_size()
    log("Training dataset has %d steps in each epoch." % steps_size)
    model = (model_factory or build_model)(config, input_dim, vocab_size, device)
    log("Total parameter of ASR model: %d." % sum(p.numel() for p in model.parameters()))
    start_epoch = 0
    if config.get("resume_ckpt"):
        from ..utils.ckpt import load_mindspore_checkpoint, read_epoch_num

        start_epoch = read_epoch_num(config["resume_ckpt"])  # train.py:121
        load_mindspore_checkpoint(model, config["resume_ckpt"], strict=False)  # (load_param_into_net does not insist either)
        log("Successfully loading the pre-trained model")
    factory = step_factory or build_step
    try:
        eng = factory(model, config, rank, world, process_group, start_steps=start_epoch * steps_size)
    except TypeError:  # (a substituted factory without the keyword)
        eng = factory(model, config, rank, world, process_group)
    max_epoch = int(config["max_epoch"])
    save_every = steps_size * int(config.get("save_checkpoint_epochs", 1))
    model_dir = os.path.join(str(config.get("exp_name", "default")), "model")
    records, step = [], start_epoch * steps_size  # (TimeMonitor.step, moved on by ResumeCallback)
    evaluator = None
    if eval_dataset is not None:
        evaluator = EvalCallback(model, eng, eval_dataset, device, world, rank, save_every, model_dir, log)
        evaluator.begin()
    log("Training start.")
    for epoch in range(start_epoch + 1, max_epoch + 1):
        # The next batch is collated (files, features, masks: host work and a little device work) AFTER this step's launches are
        # enqueued and BEFORE its results are read: the loader's ~4.7 ms hide behind the ~9.5 ms the device spends on the step
        # (round 6; `for cols in dataset: eng.step(*cols)` paid them one after the other, tools/loader_bench.py --step).  Same order of
        # random draws, same batches, same step: only the place of the host's wait moves.
        batches = iter(dataset)
        cols = next(batches, None)
        pipelined = hasattr(eng, "enqueue_step") and hasattr(eng, "finish_step")
        while cols is not None:
            t0 = time.time()
            cols = tuple(c.to(device) if hasattr(c, "to") else c for c in cols)
            if pipelined:
                pending = eng.enqueue_step(*cols)
                nxt = next(batches, None)
                loss, cond, scale, overflow, lr = eng.finish_step(*pending)
            else:
                loss, cond, scale, overflow, lr = eng.step(*cols)
                nxt = next(batches, None)
            cols = nxt
            loss, scale, lr, overflow = float(loss), float(scale), float(lr), bool(overflow)  # (the reads TimeMonitor does: a host sync)
            seconds = time.time() - t0
            log(format_step_line(epoch, max_epoch, step, steps_size, seconds, lr, loss, scale, rank, overflow))
            records.append(dict(epoch=epoch, step=step, loss=loss, scale=scale, overflow=overflow, lr=lr, seconds=seconds))
            step += 1
            if evaluator is not None:
                evaluator.step_end(epoch, max_epoch, step)
            elif config.get("save_checkpoint") and rank == 0 and step % save_every == 0 and hasattr(eng, "sync_to_module"):
                from ..utils.ckpt import write_mindspore_ckpt

                eng.sync_to_module()
                os.makedirs(model_dir, exist_ok=True)
                # ModelCheckpoint's naming (prefix CKP, epoch_step), with epoch_num appended as the reference does (train.py:157-163)
                path = os.path.join(model_dir, "CKP-%d_%d.ckpt" % (epoch, steps_size))
                from ..utils.ckpt import to_reference_names

                params = to_reference_names(model.state_dict())  # (parameters AND the BatchNorm moving statistics)
                params["epoch_num"] = __import__("numpy").asarray(epoch, dtype="int32")
                write_mindspore_ckpt(path, params)
                log("checkpoint: %s" % path)
            if max_steps is not None and step >= max_steps:
                if evaluator is not None:
                    evaluator.end()
                return records
    if evaluator is not None:
        evaluator.end()
    return records


class EvalCallback:
    """mindaudio/utils/callback.py:256-447 for the conformer script (train.py:143-155): `conformer_init.ckpt` before the first step;
    every `run_interval` steps (= steps_size x save_checkpoint_epochs) the evaluation set's utterance-weighted mean loss through
    ASREvalNet (the loss all-reduced over the ranks / device_num, asr_model.py:355-371) in evaluation mode, a log line, and
    `conformer_<epoch>_<step>.ckpt` with a `.yaml` of {loss, time} beside it; at the end the element-wise mean of the
    `num_best_ckpt` checkpoints of lowest loss as `conformer_avg_<num_best_ckpt>.ckpt` (every tensor that is not an optimizer moment:
    the BatchNorm moving statistics and epoch_num are averaged too, as the reference does).  Rank 0 writes; every rank evaluates."""

    def __init__(self, model, eng, dataset, device, world, rank, run_interval, save_ckpt_path, log, ckpt_prefix="conformer",
                 eval_log_interval=10, num_best_ckpt=30):
        from .asr_model import ASREvalNet

        self.model, self.eng, self.dataset, self.device, self.rank = model, eng, dataset, device, rank
        self.net = ASREvalNet(model, world)
        self.run_interval, self.dir, self.log = max(int(run_interval), 1), save_ckpt_path, log
        self.prefix, self.eval_log_interval, self.num_best_ckpt = ckpt_prefix, eval_log_interval, num_best_ckpt
        self.total_eval_time, self.loss_ckpt_record = 0.0, []
        if rank == 0:
            os.makedirs(save_ckpt_path, exist_ok=True)

    def evaluate(self):
        if hasattr(self.eng, "sync_to_module"):
            self.eng.sync_to_module()  # (the trained weights and BatchNorm statistics live in the engine's flat buffers)
        was_training = self.model.training
        self.model.eval()
        total_loss = total_utts = 0.0
        total_step = self.dataset.get_dataset_size()
        t0 = time.time()
        for i, cols in enumerate(self.dataset):
            cols = tuple(c.to(self.device) if hasattr(c, "to") else c for c in cols)
            loss = float(self.net(*cols))
            total_loss += loss * cols[0].shape[0]
            total_utts += cols[0].shape[0]
            if i % self.eval_log_interval == 0 and self.rank == 0:
                self.log("[EvalCallback] Step: %d/%d, Eval Loss: %.4f." % (i, total_step, loss))
        self.model.train(was_training)
        seconds = time.time() - t0
        self.total_eval_time += seconds
        return total_loss / max(total_utts, 1.0), seconds

    def save_ckpt(self, prefix, infos, epoch):
        import numpy as np
        import yaml

        from ..utils.ckpt import to_reference_names, write_mindspore_ckpt

        path = os.path.join(self.dir, prefix + ".ckpt")
        params = to_reference_names(self.model.state_dict())
        if epoch != -1:
            params["epoch_num"] = np.asarray(epoch, dtype="int32")
        write_mindspore_ckpt(path, params)
        if infos:
            with open(os.path.join(self.dir, prefix + ".yaml"), "w") as fh:
                fh.write(yaml.dump(infos))
        self.log("[EvalCallback] Successfully save %s.ckpt to %s." % (prefix, self.dir))
        return path

    def begin(self):
        if self.rank == 0:
            if hasattr(self.eng, "sync_to_module"):
                self.eng.sync_to_module()
            self.save_ckpt(self.prefix + "_init", {}, -1)

    def step_end(self, epoch, max_epoch, step):
        if step % self.run_interval != 0:
            return
        avg_loss, seconds = self.evaluate()
        if self.rank != 0:
            return
        self.log("[EvalCallback] Epoch %d/%d, Average Eval Loss: %.4f, Eval Spend Time: %dm %ds."
                 % (epoch, max_epoch, avg_loss, int(seconds // 60), int(seconds % 60)))
        path = self.save_ckpt("%s_%d_%d" % (self.prefix, epoch, step), {"loss": float(avg_loss), "time": float(seconds)}, epoch)
        self.loss_ckpt_record.append({"loss": float(avg_loss), "ckpt_path": path})

    def average_model(self):
        import numpy as np

        from ..utils.ckpt import read_mindspore_ckpt, write_mindspore_ckpt

        best = sorted(self.loss_ckpt_record, key=lambda r: r["loss"])[:self.num_best_ckpt]
        if not best:
            return None
        acc = {}
        for rec in best:
            for name, value in read_mindspore_ckpt(rec["ckpt_path"]).items():
                if not name.startswith("moment"):
                    acc.setdefault(name, []).append(np.asarray(value))
        avg = {name: np.mean(np.array(vals), axis=0).astype(vals[0].dtype) for name, vals in acc.items()}
        path = os.path.join(self.dir, "%s_avg_%d.ckpt" % (self.prefix, self.num_best_ckpt))
        write_mindspore_ckpt(path, avg)
        self.log("[EvalCallback] Successfully save %s_avg_%d.ckpt to %s." % (self.prefix, self.num_best_ckpt, self.dir))
        return path

    def end(self):
        if self.rank == 0:
            self.average_model()
            t = self.total_eval_time
            self.log("[EvalCallback] [After training] Total Eval Time: %dh %dm %ds." % (int(t // 3600), int(t % 3600 // 60), int(t % 60)))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--config_path", required=True)
    for key in ("train_data", "dict", "cmvn_file", "exp_name", "resume_ckpt"):
        ap.add_argument("--" + key)
    ap.add_argument("--max_epoch", type=int)
    ap.add_argument("--max_steps", type=int, help="stop after this many steps (smoke runs)")
    ap.add_argument("--is_distributed", type=lambda s: s.lower() in ("1", "true", "yes"))
    a = ap.parse_args(argv)
    cfg = load_config(a.config_path, {k: getattr(a, k) for k in ("train_data", "dict", "cmvn_file", "exp_name", "resume_ckpt", "max_epoch",
                                                                 "is_distributed")})
    import torch

    rank, world, pg = 0, 1, None
    if cfg.get("is_distributed"):
        import torch.distributed as dist

        rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD
    train(cfg, rank=rank, world=world, max_steps=a.max_steps, log=lambda m: print(m, flush=True), process_group=pg)
    return 0


if __name__ == "__main__":
    sys.exit(main())
