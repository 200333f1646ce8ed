#!/usr/bin/env python3
"""Golden vectors of the reference's activation functions (code/layers.py:13-80 get_act) and of its MLPBlock with
each of them (layers.py:173-188): outputs and input gradients of the REAL classes on seeded inputs, for
oracle/ref_model.act / dnn(hidden_act=) and the mapx kernels behind `--hidden_act`.  Runs only where /root/reference
exists; same import note as gen_golden.py (one attribute of transformers.utils).

    python tests/golden/gen_act_golden.py
"""
import functools
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
KINDS = ["tanh", "sigmoid", "none", "elu", "leu", "gelu", "gelu_new", "swish", "mish"]

if __name__ == "__main__":
    import transformers.utils as tu
    if not hasattr(tu, "cached_property"):
        tu.cached_property = functools.cached_property
    sys.path.insert(0, "/root/reference/code")
    import layers as L
    torch.manual_seed(7)
    x = torch.cat([torch.randn(500) * 3, torch.tensor([0.0, -0.0, 1e-6, -1e-6, 20.0, -20.0, 88.0, -88.0])])
    out = {"x": x.numpy()}
    for k in KINDS:
        xi = x.clone().requires_grad_(True)
        y = L.get_act(k)(xi)
        y.backward(torch.ones_like(y))
        out[f"{k}/y"], out[f"{k}/dy_dx"] = y.detach().numpy(), xi.grad.numpy()
    # MLPBlock(24 -> 20 x 2) with every activation, dropout 0: parameters, output, all gradients
    xin = torch.randn(9, 24)
    out["mlp/x"] = xin.numpy()
    for k in KINDS:
        torch.manual_seed(11)
        m = L.MLPBlock(24, hidden_size=20, num_hidden_layers=2, hidden_act=k, hidden_dropout_rate=0.0)
        xi = xin.clone().requires_grad_(True)
        y = m(xi)
        (y * torch.linspace(-1, 1, y.numel()).view_as(y)).sum().backward()
        for n, p in m.named_parameters():
            out[f"mlp/{k}/p/{n}"], out[f"mlp/{k}/g/{n}"] = p.detach().numpy(), p.grad.numpy()
        out[f"mlp/{k}/y"], out[f"mlp/{k}/dx"] = y.detach().numpy(), xi.grad.numpy()
    path = os.path.join(HERE, "activations.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")
