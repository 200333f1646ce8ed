"""How the two-piece fp16 arithmetic (csrc/gemm_h2.hip) degrades when ONE element of operand A is 2^k times the
rest (per-tensor scaling): worst error of the OTHER rows relative to sum_k |a_k b_k|, per k.
    python tools/micro/h2_range_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "map-code_amd"))
from mapx import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
M, N, K = 256, 128, 512
for k in (0, 4, 8, 12, 14, 16, 18, 20, 24, 28, 32, 40, 50):
    A = torch.randn(M, K, generator=g)
    A[3, 7] = 2.0 ** k
    B = torch.randn(N, K, generator=g)
    Ad, Bd = A.cuda(), B.cuda()
    out = ops.gemm(Ad, Bd, True, True, M, N, K, amax_a=ops.amax(Ad), amax_b=ops.amax(Bd)).cpu()
    ref, mag = A.double() @ B.double().t(), A.abs().double() @ B.abs().double().t()
    err = ((out.double() - ref).abs() / mag)
    keep = torch.ones(M, dtype=torch.bool)
    keep[3] = False
    print(f"outlier 2^{k:2d}: other rows max err/mag {float(err[keep].max()):.2e}  rms {float((err[keep] ** 2).mean().sqrt()):.2e}   outlier row {float(err[3].max()):.2e}")
