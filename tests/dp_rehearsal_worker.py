"""Worker for test_one_rank_rccl_exchange_equals_plain_step: 2 epochs of MFP pretraining through
mapx.trainer.Trainer, optionally behind a ONE-RANK RCCL group (MAPX_FORCE_DP=1: the step then
runs every collective and kernel of the N-rank gradient exchange), graph or eager; saves the
parameters."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


def main(out, use_graph):
    forced = os.environ.get("MAPX_FORCE_DP", "0") == "1"
    if forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        torch.cuda.set_device(0)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        from mapx import parallel
        parallel.init_rccl(torch.device("cuda:0"))
    from mapx.arguments import TrainingArguments
    from mapx.dataset import OurDataset, synth_table
    from mapx.models import BaseModel
    from mapx.trainer import Trainer
    from util import make_config
    cfg = dict(F=23, V=3000, E=16, H=64, NL=3, NC=3, P=32, K=25)
    ids, labels, _, _ = synth_table(512 * 9 + 100, 23, cfg["V"], seed=3)       # ragged last batch
    cnt = np.bincount(ids.reshape(-1), minlength=cfg["V"]).astype(np.float32)
    torch.manual_seed(5)
    config = make_config(cfg, "MFP", cnt)
    model = BaseModel.from_config(config)
    targs = TrainingArguments(output_dir=os.path.dirname(out), per_gpu_train_batch_size=512,
                              per_gpu_eval_batch_size=512, learning_rate=1e-3, lr_sched="cosine",
                              weight_decay=5e-2, num_train_epochs=2, pretrain=True, pt_type="MFP",
                              sampling_method="randint", mask_ratio=0.3, logging_steps=7, seed=11)
    targs._device = torch.device("cuda:0")
    tr = Trainer(model, config, targs, OurDataset(ids, labels), OurDataset(ids[:600], labels[:600]))
    tr.use_graph = use_graph
    tr.MFP_pretrain()
    assert tr.global_step == 2 * 10
    kinds = [type(g).__name__ for g in tr._graphs.values() if not isinstance(g, int)]
    assert kinds == ((["GraphedBackward"] if forced else ["GraphedStep"]) if use_graph else []), kinds
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, out)
    torch.cuda.synchronize()
    if forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] == "graph")
