"""Helpers shared by the parity tests: golden loading and digest comparison."""
import os

import numpy as np
import torch

import paramgen as pg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = ("MFP", "RFD", "CTR")


def load_case(case, mode, backbone="DCNv2"):
    cfg = pg.CASES[case]
    suffix = "" if backbone == "DCNv2" else f"_{backbone}"
    z = np.load(os.path.join(GOLD, f"{case}_{mode}{suffix}.npz"))
    inp = pg.make_inputs(case, cfg)
    params = pg.make_params(case, cfg, mode, backbone)
    return cfg, z, inp, params


def t(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def assert_digest(z, prefix, name, got, rtol=2e-5, atol=1e-6):
    """Compare `got` with what the fixture kept of tensor `name` (see paramgen.digest)."""
    got = np.asarray(got, dtype=np.float32)
    want = {k.split("/")[-1]: z[k] for k in z.files if k.startswith(f"{prefix}/{name}/")}
    assert want, f"no golden entry for {prefix}/{name}"
    mine = pg.digest(name, got)
    assert set(mine) == set(want), (name, set(mine), set(want))
    for k, w in want.items():
        g = np.asarray(mine[k], dtype=np.float64)
        w = np.asarray(w, dtype=np.float64)
        scale = max(1.0, float(np.abs(w).max())) if k.startswith("proj") or k == "full" else 1.0
        if k in ("proj_r", "proj_l", "sum"):
            # linear functionals of ~N(0, s) entries: tolerance relative to the abs-sum scale
            scale = max(scale, float(np.asarray(want.get("abssum", 1.0))) * 1e-3)
        np.testing.assert_allclose(g, w, rtol=rtol, atol=atol * scale,
                                   err_msg=f"{prefix}/{name}/{k}")


def make_config(cfg, mode, feat_count=None, data_dir=None, seed=42, backbone="DCNv2", compute_dtype="fp32"):
    """mapx Config for a fixture case (the 11 runtime keys of reference run.py:50-61 + flags)."""
    from mapx.arguments import Config
    extra = pg.extras_of(backbone)
    return Config(**extra, compute_dtype=compute_dtype, model_name=backbone, data_dir=data_dir, input_size=cfg["V"], num_fields=cfg["F"],
                  embed_size=cfg["E"], embed_dropout_rate=0.0, embed_norm=False, layer_norm_eps=1e-12,
                  hidden_size=cfg["H"], num_hidden_layers=cfg["NL"], hidden_act="relu",
                  hidden_dropout_rate=0.0, num_cross_layers=cfg["NC"], pt_neg_num=cfg["K"],
                  proj_size=cfg["P"], pretrain=(mode != "CTR"),
                  pt_type=("RFD" if mode == "RFD" else "MFP"), RFD_replace="Unigram",
                  feat_count=None if feat_count is None else torch.as_tensor(feat_count),
                  device=None, n_gpu=1, idx_low=None, idx_high=None, feat_num_per_field=None, seed=seed)


def build_model(cfg, mode, params, feat_count, device="cuda", backbone="DCNv2", compute_dtype="fp32"):
    """The backbone with the fixture's reproducible parameters loaded."""
    from mapx.models import BaseModel
    model = BaseModel.from_config(make_config(cfg, mode, feat_count, backbone=backbone, compute_dtype=compute_dtype))
    with torch.no_grad():
        sd = model.state_dict()
        for k, v in params.items():
            sd[k].copy_(torch.from_numpy(v))
    return model.to(device)
