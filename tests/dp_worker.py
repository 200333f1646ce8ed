"""Worker for test_data_parallel_two_ranks_equal_single_process: runs 4 MFP steps on its row
shard (WORLD_SIZE ranks, gloo) or on the whole batch (no launcher) and saves the parameters."""
import os
import sys

import torch
import torch.distributed as dist

import paramgen as pg
from util import build_model, load_case, t


def main(out):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo")
    from mapx import ops, parallel
    from mapx.arguments import TrainingArguments
    from mapx.optim import MapxOptimizer
    case = "B_f25_b64"
    cfg = pg.CASES[case]
    _, _, inp, params = load_case(case, "MFP")
    model = build_model(cfg, "MFP", params, inp["feat_count"])
    targs = TrainingArguments(output_dir="/tmp/x", learning_rate=1e-3, weight_decay=5e-2, lr_sched="cosine")
    opt = MapxOptimizer(model, targs, num_training_steps=8, num_warmup_steps=0)
    B = cfg["B"]
    lo, hi = (rank * B // world, (rank + 1) * B // world)
    L = inp["masked_index"].shape[1]
    model.train()
    for step in range(4):
        g = torch.Generator().manual_seed(step)
        perm = torch.randperm(B, generator=g)
        ids = t(inp["input_ids"])[perm][lo:hi].to("cuda")
        mi = t(inp["masked_index"])[perm][lo:hi].to("cuda")
        noise = t(inp["noise"])[perm][lo:hi].to("cuda")
        masked, labels, _ = ops.dynamic_mask_mfp(ids, L, masked_index=mi)
        loss = model(input_ids=masked, labels=labels, masked_index=mi, noise_samples=noise)[0]
        loss.backward()
        parallel.sync_gradients(opt)
        opt.step()
    opt.flush()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items() if v.dtype == torch.float32}
    torch.save(sd, out if world == 1 else f"{out}.{rank}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
