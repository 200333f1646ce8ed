"""Argument surface of the reference's `run.py` (code/arguments.py:15-161, parsed there by
HfArgumentParser at run.py:14-15), rebuilt without the transformers dependency.

Flag names, types and defaults are pinned against the reference's dataclasses by
tests/test_host_surface.py (golden: tests/golden/flag_surface.json).  Every flag accepts
`--name=value` and `--name value`; booleans also accept the bare form (`--finetune`) and
`--pretrain=True/False`, as the reference's run scripts use them.
"""
import argparse
import copy
import json
import os

import torch

REQUIRED = object()

# (name, type, default, help) — the training / data / pretraining flags
TRAINING_FLAGS = [
    ("output_dir", str, REQUIRED, "directory for train.log, results.log and {step}.model checkpoints"),
    ("dataset_name", str, "avazu", "prefix of <name>-meta.json / <name>.h5 inside data_dir"),
    ("data_dir", str, "data/avazu", "directory holding the preprocessed dataset"),
    ("per_gpu_train_batch_size", int, 128, "rows per GPU per training step"),
    ("per_gpu_eval_batch_size", int, 10000, "rows per GPU per evaluation step"),
    ("learning_rate", float, 1e-4, "peak AdamW learning rate"),
    ("weight_decay", float, 0.1, "decoupled weight decay (not applied to biases / LayerNorm.weight)"),
    ("adam_epsilon", float, 1e-8, "AdamW epsilon"),
    ("adam_betas", str, "0.9,0.999", "AdamW beta1,beta2"),
    ("max_grad_norm", float, 0.0, "global-norm gradient clipping; 0 disables"),
    ("patience", int, 2, "early-stopping patience (evaluations without AUC improvement)"),
    ("num_train_epochs", int, 20, "number of passes over the training split"),
    ("lr_sched", str, "cosine", "cosine | const (both with linear warmup)"),
    ("warmup_ratio", float, 0.0, "fraction of total steps used for linear warmup"),
    ("logging_first_step", bool, False, "accepted for compatibility; unused (as in the reference)"),
    ("logging_steps", int, 1000, "log window metrics every N optimizer steps"),
    ("save_steps", int, 1000, "accepted for compatibility; unused (as in the reference)"),
    ("save_total_limit", "optional_int", 20, "accepted for compatibility; unused (as in the reference)"),
    ("no_cuda", bool, False, "refuse the GPU (mapx has no CPU path: raises)"),
    ("seed", int, 42, "seed for parameter init, shuffling, masks and negative sampling"),
    ("local_rank", int, -1, "rank inside the node; set from LOCAL_RANK under torchrun"),
    ("sampling_method", str, "normal", "masked-field sampling: normal (no replacement) | randint"),
    ("mask_ratio", float, 0.1, "fraction of fields masked / replaced per row"),
    ("pretrain", bool, False, "run self-supervised pretraining instead of CTR training"),
    ("pt_type", str, "MFP", "pretraining task: MFP | RFD"),
    ("RFD_replace", str, "Unigram", "RFD replacement generator: Unigram | Uniform | Whole-Uniform | Whole-Unigram"),
    ("finetune", bool, False, "initialise from --pretrained_model_path before CTR training"),
    ("pretrained_model_path", str, None, "checkpoint ({step}.model) to finetune from"),
]

# the model flags (only a subset drives DCNv2; the rest is accepted so that any reference
# command line parses unchanged)
MODEL_FLAGS = [
    ("model_name", str, REQUIRED, "backbone; this build implements DCNv2"),
    ("embed_size", int, 32, "embedding width E"),
    ("embed_dropout_rate", float, 0.0, ""),
    ("hidden_size", int, 128, "deep-tower width H"),
    ("num_hidden_layers", int, 1, "deep-tower depth"),
    ("hidden_act", str, "relu", ""),
    ("hidden_dropout_rate", float, 0.0, ""),
    ("num_attn_heads", int, 1, ""), ("attn_probs_dropout_rate", float, 0.1, ""),
    ("intermediate_size", int, 128, ""), ("norm_first", bool, False, ""),
    ("layer_norm_eps", float, 1e-12, ""), ("agg_type", str, "mean", ""),
    ("res_conn", bool, False, ""), ("num_channels", int, 1, ""),
    ("embed_norm", bool, False, ""), ("prod_layer_norm", bool, False, ""),
    ("prod_dropout_rate", float, 0.1, ""), ("inter_layer_norm", bool, False, ""),
    ("output_reduction", str, "sum,max,sum", ""),
    ("num_cross_layers", int, 1, "number of CrossNetV2 layers"),
    ("share_embedding", bool, False, ""), ("channels", str, "14,16,18,20", ""),
    ("kernel_heights", str, "7,7,7,7", ""), ("pooling_sizes", str, "2,2,2,2", ""),
    ("recombined_channels", str, "3,3,3,3", ""), ("conv_act", str, "tanh", ""),
    ("reduction_ratio", int, 3, ""), ("bilinear_type", str, "field_interaction", ""),
    ("reuse_graph_layer", bool, False, ""), ("attn_scale", bool, False, ""),
    ("use_lr", bool, False, ""), ("attn_size", int, 40, ""), ("num_attn_layers", int, 2, ""),
    ("cin_layer_units", str, "50,50", ""), ("field_interaction_type", str, "matrixed", ""),
    ("product_type", str, "inner", ""), ("outer_product_kernel_type", str, "mat", ""),
    ("pt_neg_num", int, 25, "NCE negatives per masked feature (K)"),
    ("proj_size", int, 32, "per-field projection width P of the pretraining heads"),
    ("dnn_size", int, 1000, ""), ("num_dnn_layers", int, 0, ""),
    ("dnn_act", str, "relu", ""), ("dnn_drop", float, 0.0, ""),
]


# mapx extensions: flags the reference does not have (kept apart so that the reference surface
# above stays pinned one-to-one by tests/golden/flag_surface.json)
EXTENSION_MODEL_FLAGS = [
    ("compute_dtype", str, "fp32", "fp32 | bf16: dtype of the trunk's activations and GEMM operands "
                                   "(bf16: fp32 master weights, fp32 accumulation, fp32 tables and optimizer)"),
]


def _to_bool(v):
    if isinstance(v, bool):
        return v
    s = str(v).strip().lower()
    if s in ("true", "1", "yes", "y", "t"):
        return True
    if s in ("false", "0", "no", "n", "f"):
        return False
    raise argparse.ArgumentTypeError(f"not a boolean: {v!r}")


def _optional_int(v):
    return None if str(v).lower() in ("none", "") else int(v)


class _Bag:
    _flags = ()

    def __init__(self, **kw):
        for name, _typ, default, _help in self._flags:
            if name in kw:
                val = kw.pop(name)
            elif default is REQUIRED:
                raise TypeError(f"{type(self).__name__} missing required argument {name!r}")
            else:
                val = default
            setattr(self, name, val)
        if kw:
            raise TypeError(f"unknown arguments {sorted(kw)}")

    def to_dict(self):
        return copy.deepcopy({n: getattr(self, n) for n, *_ in self._flags})

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2)

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{k}={v!r}' for k, v in self.to_dict().items())})"


class ModelArguments(_Bag):
    _flags = MODEL_FLAGS + EXTENSION_MODEL_FLAGS


class TrainingArguments(_Bag):
    """Device policy differs from the reference on purpose (the reference only calls
    init_process_group and then trains unsynchronised replicas).  Here one process drives one
    GPU; under `torchrun` the process group is RCCL ("nccl") and `mapx.parallel` keeps the
    replicas identical."""
    _flags = TRAINING_FLAGS
    _device = None

    @property
    def world_size(self):
        return int(os.environ.get("WORLD_SIZE", "1"))

    @property
    def device(self):
        if self._device is None:
            if self.no_cuda or not torch.cuda.is_available():
                raise RuntimeError("mapx runs on an MI355X only (no CPU path); --no_cuda is not supported")
            lr = int(os.environ.get("LOCAL_RANK", "-1"))
            if self.world_size > 1:
                self.local_rank = int(os.environ.get("RANK", "0"))
                torch.cuda.set_device(max(lr, 0))
                if not torch.distributed.is_initialized():
                    from . import parallel
                    parallel.init_rccl()
            self._device = torch.device("cuda", max(lr, 0))
        return self._device

    @property
    def n_gpu(self):
        return max(1, self.world_size)

    @property
    def train_batch_size(self):
        return self.per_gpu_train_batch_size * max(1, self.n_gpu)

    @property
    def eval_batch_size(self):
        return self.per_gpu_eval_batch_size * max(1, self.n_gpu)


def build_parser():
    ap = argparse.ArgumentParser(prog="run.py", allow_abbrev=False,
                                 description="DCNv2 scratch / MFP / RFD / finetune on MI355X")
    for name, typ, default, hlp in MODEL_FLAGS + EXTENSION_MODEL_FLAGS + TRAINING_FLAGS:
        kw = dict(dest=name, help=hlp or None)
        if default is REQUIRED:
            kw["required"] = True
        else:
            kw["default"] = default
        if typ is bool:
            kw.update(type=_to_bool, nargs="?", const=True)
        elif typ == "optional_int":
            kw.update(type=_optional_int)
        else:
            kw.update(type=typ)
        ap.add_argument(f"--{name}", **kw)
    return ap


def parse_args_into_dataclasses(argv=None):
    """-> (ModelArguments, TrainingArguments), like HfArgumentParser((ModelArguments,
    TrainingArguments)).parse_args_into_dataclasses() in reference run.py:14-15."""
    ns = vars(build_parser().parse_args(argv))
    margs = ModelArguments(**{n: ns[n] for n, *_ in MODEL_FLAGS + EXTENSION_MODEL_FLAGS})
    targs = TrainingArguments(**{n: ns[n] for n, *_ in TRAINING_FLAGS})
    return margs, targs


class Config:
    """Attribute bag + json round trip (reference arguments.py:164-203)."""

    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)

    @classmethod
    def from_dict(cls, d):
        return cls(**d)

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def _jsonable(self):
        out = {}
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor):
                continue                      # feat_count, idx_low ... are runtime tensors
            out[k] = str(v) if isinstance(v, torch.device) else v
        return out

    def to_json_string(self):
        return json.dumps(self._jsonable(), indent=2, sort_keys=True) + "\n"

    def save(self, save_directory):
        assert os.path.isdir(save_directory), f"not a directory: {save_directory}"
        with open(os.path.join(save_directory, "config.json"), "w", encoding="utf-8") as f:
            f.write(self.to_json_string())

    @classmethod
    def load(cls, load_directory):
        with open(os.path.join(load_directory, "config.json"), "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))
