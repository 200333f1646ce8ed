"""GPU time of tall weight gradients with both dimensions small (AutoInt's attention projections: dW [40, 16 | 40] over
B*F = 94 208 rows): the streaming kernel (csrc/skinny.hip, skinny_dw_tall_kernel) against the one-tile split-K GEMM.
    python tools/micro/tall_bench.py        (MAPX_TALL_ROWS=<rows per chunk> sweeps the chunking)"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "map-code_amd"))
sys.path.insert(0, os.path.join(HERE, ".."))
from mapx import ops  # noqa: E402
from gemm_f32_bench import timeit  # noqa: E402

if __name__ == "__main__":
    for M, N, K in [(94208, 40, 16), (94208, 40, 40)]:
        x = torch.randn(M, K, device="cuda")
        dy = torch.randn(M, N, device="cuda")
        w = torch.randn(N, K, device="cuda")
        b = torch.randn(N, device="cuda")
        line = f"{M}x{N}x{K}:"
        for sk in (True, False):
            ops.SKINNY = sk
            dw = timeit(lambda: ops.linear_bwd_weight(dy, x))
            dx = timeit(lambda: ops.linear_bwd_input(dy, w))
            f = timeit(lambda: ops.linear_fwd(x, w, b))
            line += f"  {'skinny' if sk else 'gemm'} dW {dw:6.1f} dX {dx:6.1f} fwd {f:6.1f} |"
        print(line)
