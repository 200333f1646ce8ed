"""How many of a step's fp32 products are handed both magnitude records (and so take csrc/gemm_h2.hip), per step kind:
    python tools/h2_usage.py        (MAPX_AMAX_CHECK=1: every record is also compared with its operand on the host)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
from mapx import ops  # noqa: E402
from mapx.arguments import TrainingArguments  # noqa: E402
from mapx.dataset import OurDataset, synth_table  # noqa: E402
from mapx.models import BaseModel  # noqa: E402
from mapx.trainer import Trainer  # noqa: E402
from util import make_config  # noqa: E402

for model_name in ("DCNv2", "DNN"):
    for pt in ("MFP", "RFD", "CTR"):
        cfg = dict(F=23, V=3000, E=16, H=256, NL=3, NC=3, P=32, K=25)
        ids, labels, _, _ = synth_table(512 * 6, 23, cfg["V"], seed=3)
        cnt = np.bincount(ids.reshape(-1), minlength=cfg["V"]).astype(np.float32)
        torch.manual_seed(5)
        config = make_config(cfg, pt, cnt, backbone=model_name)
        model = BaseModel.from_config(config)
        targs = TrainingArguments(output_dir="/tmp/h2u", per_gpu_train_batch_size=512, per_gpu_eval_batch_size=512,
                                  learning_rate=1e-3, lr_sched="cosine", weight_decay=5e-2, num_train_epochs=1,
                                  pretrain=pt != "CTR", pt_type=pt if pt != "CTR" else "MFP", sampling_method="randint", RFD_replace="Unigram",
                                  mask_ratio=0.3, seed=11)
        targs._device = torch.device("cuda")
        ds = OurDataset(ids, labels)
        tr = Trainer(model, config, targs, ds, ds)
        tr.use_graph = False
        train = tr._begin("test")
        model.train()
        kind = {"MFP": "mfp", "RFD": "rfd", "CTR": "ctr"}[pt]
        for i, (X, Y) in enumerate(train.batches(512, True, tr._generator(), (0, 1))):
            ops.H2_USED[0] = ops.H2_USED[1] = 0
            tr.run_step(kind, X, Y)
            if i == 2:
                break
        torch.cuda.synchronize()
        print(f"{model_name:6s} {pt}: products with both records {ops.H2_USED[0]:3d}, without {ops.H2_USED[1]:3d}", flush=True)
