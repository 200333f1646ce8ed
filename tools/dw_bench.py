#!/usr/bin/env python3
"""Time the weight-gradient GEMM dW = dY^T X on the step's shapes and check it against a
float64 product."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "map-code_amd"))
import torch
from mapx import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, N, K in ((4096, 1000, 1000), (4096, 1000, 368), (4096, 368, 368), (4096, 736, 1368), (512, 64, 368)):
    dy, x = torch.randn(B, N, device=dev), torch.randn(B, K, device=dev)
    out = ops.linear_bwd_weight(dy, x)
    ref = dy.double().t() @ x.double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    for _ in range(5):
        ops.linear_bwd_weight(dy, x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(2_000_000)
    e0.record()
    for _ in range(50):
        ops.linear_bwd_weight(dy, x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"dW [{N}x{K}] over {B}: {us:7.1f} us  {2.0 * B * N * K / us / 1e6:6.1f} TF  rel.err {err:.1e}")
