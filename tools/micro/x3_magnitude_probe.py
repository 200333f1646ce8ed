"""What the six-product fp32 GEMM (csrc/gemm_x3.hip) does with operands far from randn: exponent spread inside a
row, rows near the ends of the fp32 range, subnormal values, values above the largest bf16, infinities and NaNs.
Prints the worst error against fp64 relative to sum_k |a_k b_k| per case; tests/test_kernels_gpu.py pins what it shows.
    python tools/micro/x3_magnitude_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "map-code_amd"))
from mapx import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
M, N, K = 256, 128, 512


def spread(shape, lo, hi):
    mant = 1 + torch.rand(shape, generator=g)
    e = torch.randint(lo, hi + 1, shape, generator=g).float()
    sign = torch.where(torch.rand(shape, generator=g) < 0.5, -1.0, 1.0)
    return sign * mant * torch.exp2(e)


def run(name, A, B):
    out = ops.gemm(A.cuda(), B.cuda(), True, True, M, N, K).cpu()
    ref = A.double() @ B.double().t()
    mag = A.abs().double() @ B.abs().double().t()
    fin = torch.isfinite(ref) & torch.isfinite(mag) & (mag > 0)
    rel = ((out.double() - ref).abs() / mag)[fin]
    print(f"{name:44s} worst |err| / sum|ab| = {float(rel.max()) if rel.numel() else float('nan'):.3e}   "
          f"non-finite outputs {int((~torch.isfinite(out)).sum())} (reference {int((~torch.isfinite(ref.float())).sum())})")
    return out


run("randn", torch.randn(M, K, generator=g), torch.randn(N, K, generator=g))
run("exponents 2^-30..2^30 inside a row", spread((M, K), -30, 30), spread((N, K), -30, 30))
run("A rows x 2^100, B x 2^-100", spread((M, K), -3, 3) * 2.0 ** 100, spread((N, K), -3, 3) * 2.0 ** -100)
run("A near fp32 max / 4, B ~ 1e-38", spread((M, K), 0, 0) * 2.0 ** 124, spread((N, K), 0, 0) * 2.0 ** -125)
sub = spread((M, K), 0, 0) * 2.0 ** -130          # subnormal fp32 (min normal 2^-126)
run("A subnormal (2^-130), B ~ 2^20", sub, spread((N, K), 20, 20))
run("A 2^-120, B 2^-20 (products subnormal)", spread((M, K), -120, -120), spread((N, K), -20, -20))
A = torch.randn(M, K, generator=g)
A[3, 7] = 3.40e38                                  # above the largest bf16 (3.3895e38), below fp32 max
B = torch.randn(N, K, generator=g) * 1e-3
o = run("one element 3.40e38 (> bf16 max)", A, B)
print("   row 3 finite:", bool(torch.isfinite(o[3]).all()), " other rows finite:", bool(torch.isfinite(o[:3]).all() and torch.isfinite(o[4:]).all()))
A = torch.randn(M, K, generator=g); A[5, 1] = float("inf"); A[6, 2] = float("nan")
o = run("one +inf, one NaN", A, torch.randn(N, K, generator=g))
print("   row 5 (inf): all non-finite", bool((~torch.isfinite(o[5])).all()), " NaN count", int(torch.isnan(o[5]).sum()),
      "| row 6 (NaN): all NaN", bool(torch.isnan(o[6]).all()), "| other rows finite", bool(torch.isfinite(o[:5]).all() and torch.isfinite(o[7:]).all()))
