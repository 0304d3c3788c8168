"""Child process of tests/test_conformer_train_script.py (not a test module): one data-parallel rank of the training SCRIPT.

  python tests/train_worker.py <rank> <world> <port> <workdir> <out.pt>

All ranks share GPU 0 and talk over gloo on device tensors (as tests/dp_worker.py); <workdir> holds train.csv, lang_char.txt and
conformer.yaml.  Runs mindaudio_amd.conformer.train.train() with the real dataset / model / step factories and saves the logged losses
and the engine's final float32 masters."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, work, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import numpy as np
    import torch

    torch.cuda.set_device(0)
    pg = None
    if world > 1:
        import torch.distributed as dist

        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD
    from mindaudio_amd.conformer import train as T

    cfg = T.load_config(os.path.join(work, "conformer.yaml"), dict(train_data=os.path.join(work, "train.csv"),
                                                                  dict=os.path.join(work, "lang_char.txt"), max_epoch=2,
                                                                  exp_name=os.path.join(work, "exp"), is_distributed=world > 1))
    cfg["dataset_conf"]["batch_bucket_limit"] = "4, 4, 4, 4, 4, 4, 4, 4, 4, 4"
    cfg["collate_conf"].update(use_speed_perturb=True)
    cfg["scheduler_conf"]["warmup_steps"] = 5
    kept = {}

    def step_factory(model, config, r, w, process_group=None, start_steps=0):
        kept["eng"] = T.build_step(model, config, r, w, process_group, start_steps=start_steps)
        return kept["eng"]

    random.seed(11)  # every rank draws the same batch order and the same augmentation decisions (dataset.py:552-553, 693)
    np.random.seed(11)
    recs = T.train(cfg, rank=rank, world=world, log=lambda _l: None, step_factory=step_factory, process_group=pg)
    torch.cuda.synchronize()
    torch.save({"losses": [r["loss"] for r in recs], "master": kept["eng"].fp.master.cpu(),
                "bn": [m.cpu() for m in kept["eng"].bn_mean]}, out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
