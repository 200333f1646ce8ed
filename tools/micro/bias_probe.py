"""Mean SIGNED error (in units of eps32 * |C|) of the GEMM kernels on same-sign data: tells rounding
(unbiased) from chopping toward zero (error sign = -sign(C)) from flooring (negative for both signs)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "map-code_amd"))
import torch
from mapx import ops
EPS = 2.0 ** -24
g = torch.Generator().manual_seed(0)
for K in (16, 64, 368, 4096):
    M, N = 512, 512
    A = torch.rand(M, K, generator=g) + 0.5
    B = torch.rand(N, K, generator=g) + 0.5
    for sgn in (1.0, -1.0):
        Bs = B * sgn
        ref = A.double() @ Bs.double().t()
        out = ops.gemm(A.cuda(), Bs.cuda(), True, True, M, N, K).double().cpu()
        rel = ((out - ref) / ref.abs()) / EPS
        # bf16-exact operands through the plain bf16 kernel (products exact: only the accumulation rounds)
        Ab, Bb = A.to(torch.bfloat16), Bs.to(torch.bfloat16)
        refb = Ab.double() @ Bb.double().t()
        outb = ops.gemm_bf16(Ab.cuda(), Bb.cuda(), True, True, M, N, K, out_dtype=torch.float32).double().cpu()
        relb = ((outb - refb) / refb.abs()) / EPS
        print(f"{os.environ.get('MAPX_GEMM','x3'):7s} K={K:5d} C{'>' if sgn > 0 else '<'}0: f32 entry mean err {float(rel.mean()):+7.3f} u "
              f"(rms {float(rel.pow(2).mean().sqrt()):6.3f});  bf16 MFMA on bf16 data: mean {float(relb.mean()):+7.3f} u (rms {float(relb.pow(2).mean().sqrt()):6.3f})")
