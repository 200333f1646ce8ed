#!/usr/bin/env python3
"""Entry script with the reference's argument surface and output layout (code/run.py):

    python map-code_amd/run.py --model_name=DCNv2 --output_dir=... [--pretrain=True --pt_type=MFP ...]

Writes output_dir/train.log, output_dir/{global_step}.model, and copies train.log to
results.log when the job finishes (an existing results.log means "already done": exit 0).
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N map-code_amd/run.py ...`.
"""
import logging
import os
import random
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")     # before the HIP runtime starts; see mapx/__init__.py

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from mapx.arguments import Config, parse_args_into_dataclasses  # noqa: E402
from mapx.dataset import BaseDataset  # noqa: E402
from mapx.models import BaseModel  # noqa: E402
from mapx.trainer import Trainer  # noqa: E402


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None):
    model_args, training_args = parse_args_into_dataclasses(argv)
    os.makedirs(training_args.output_dir, exist_ok=True)
    training_log = os.path.join(training_args.output_dir, "train.log")
    results_log = os.path.join(training_args.output_dir, "results.log")
    if os.path.exists(results_log):
        print("job already finished, quit")
        return 0
    device = training_args.device                      # initialises RCCL under torchrun
    logging.basicConfig(format="%(message)s",
                        level=logging.INFO if training_args.local_rank in [-1, 0] else logging.WARN,
                        force=True)
    logger = logging.getLogger()
    if training_args.local_rank in [-1, 0]:
        logger.addHandler(logging.FileHandler(filename=training_log, mode="w"))
    logger.warning(f"Process rank: {training_args.local_rank}, device: {device}, n_gpu: {training_args.n_gpu}, "
                   f"distributed_training: {bool(training_args.local_rank != -1)}")
    logger.info(f"training/evaluation parameters {training_args}")
    set_seed(training_args.seed)

    dataset = BaseDataset(training_args)
    datasets = {split: dataset.get_splited_dataset(split) for split in dataset.split_names}
    logger.info(f"field_names = {dataset.field_names}")

    cfg = model_args.to_dict()
    cfg.update(data_dir=training_args.data_dir, input_size=len(dataset.feat_map),
               num_fields=len(dataset.field_map) - 1,   # field_map carries one reserved entry
               pretrain=training_args.pretrain, pt_type=training_args.pt_type,
               RFD_replace=training_args.RFD_replace, feat_count=dataset.feat_count, device=device,
               n_gpu=training_args.n_gpu, idx_low=dataset.idx_low, idx_high=dataset.idx_high,
               feat_num_per_field=dataset.feat_num_per_field, seed=training_args.seed,
               rank=max(training_args.local_rank, 0))
    config = Config.from_dict(cfg)
    model = BaseModel.from_config(config)
    if training_args.finetune:
        model.load_for_finetune(training_args.pretrained_model_path)

    trainer = Trainer(model, config, training_args, train_dataset=datasets["train"],
                      eval_dataset=datasets["valid"])
    if training_args.pretrain:
        if training_args.pt_type == "MFP":
            trainer.MFP_pretrain()
        elif training_args.pt_type == "RFD":
            trainer.RFD_pretrain()
        else:
            raise NotImplementedError
    else:
        trainer.train()
        trainer.test(datasets["test"])

    if training_args.local_rank in [-1, 0]:
        with open(training_log, "r") as src, open(results_log, "w") as dst:
            dst.write(src.read())
    return 0


if __name__ == "__main__":
    sys.exit(main())
