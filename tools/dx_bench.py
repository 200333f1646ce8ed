#!/usr/bin/env python3
"""Time the input-gradient GEMM dX = dY W on the step's shapes and check it against float64."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "map-code_amd"))
import torch
from mapx import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
for M, N, K in ((4096, 1368, 736), (4096, 1000, 1000), (4096, 368, 1000), (4096, 368, 368)):
    dy, w = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    out = ops.linear_bwd_input(dy, w)
    ref = dy.double() @ w.double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    for _ in range(5):
        ops.linear_bwd_input(dy, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(2_000_000)
    e0.record()
    for _ in range(50):
        ops.linear_bwd_input(dy, w)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"dX [{M}x{N}] K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF  rel.err {err:.1e}")
