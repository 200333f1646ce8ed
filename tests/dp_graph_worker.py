"""Worker for test_two_ranks_graphed_exchange_equals_eager: WORLD_SIZE gloo ranks on ONE GPU run MFP pretraining
through mapx.trainer.Trainer, every rank on its own rows with its own masks and negatives (rank-offset Philox
streams), with the step either eager or replayed from the graphs of the data-parallel path (GraphedBackward +
GraphedExchangeTail, MAPX_DP_GLOO_GRAPH=1); saves the rank's parameters."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist


def main(out, use_graph):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from mapx.arguments import TrainingArguments
    from mapx.dataset import OurDataset, synth_table
    from mapx.models import BaseModel
    from mapx.trainer import Trainer
    from util import make_config
    cfg = dict(F=23, V=3000, E=16, H=64, NL=3, NC=3, P=32, K=25)
    ids, labels, _, _ = synth_table(256 * 2 * 7 + 50, 23, cfg["V"], seed=3)       # 7 rounds of 2 x 256 rows + a ragged tail
    cnt = np.bincount(ids.reshape(-1), minlength=cfg["V"]).astype(np.float32)
    torch.manual_seed(5)
    config = make_config(cfg, "MFP", cnt)
    config.rank = rank
    model = BaseModel.from_config(config)
    targs = TrainingArguments(output_dir=os.path.dirname(out), per_gpu_train_batch_size=256,
                              per_gpu_eval_batch_size=256, learning_rate=1e-3, lr_sched="cosine",
                              weight_decay=5e-2, num_train_epochs=2, pretrain=True, pt_type="MFP",
                              sampling_method="randint", mask_ratio=0.3, logging_steps=100, seed=11)
    targs._device = torch.device("cuda:0")
    tr = Trainer(model, config, targs, OurDataset(ids, labels), OurDataset(ids[:300], labels[:300]))
    assert tr.world == world and tr.rank == rank
    tr.use_graph = use_graph
    tr.MFP_pretrain()
    assert tr.global_step == 2 * 7, tr.global_step
    kinds = [type(g).__name__ for g in tr._graphs.values() if not isinstance(g, int)]
    assert kinds == (["GraphedBackward"] if use_graph else []), kinds
    if use_graph:
        gb = next(g for g in tr._graphs.values() if not isinstance(g, int))
        assert gb.early and gb.tails, "the captured exchange tail did not run"
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, f"{out}.{rank}")
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] == "graph")
