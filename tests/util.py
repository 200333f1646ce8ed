"""Helpers shared by the parity tests: golden loading and digest comparison."""
import os

import numpy as np
import torch

import paramgen as pg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MODES = ("MFP", "RFD", "CTR")


def load_case(case, mode):
    cfg = pg.CASES[case]
    z = np.load(os.path.join(GOLD, f"{case}_{mode}.npz"))
    inp = pg.make_inputs(case, cfg)
    params = pg.make_params(case, cfg, mode)
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
