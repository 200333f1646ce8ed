"""GPU time of the heads' narrow layers, streaming kernels (csrc/skinny.hip) against the MFMA GEMM path:
    python tools/skinny_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
from mapx import ops  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_f32_bench import timeit  # noqa: E402

if __name__ == "__main__":
    for M, N, K in [(4096, 23, 736), (4096, 1, 1368), (4096, 39, 1248)]:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        dy = torch.randn(M, N, device="cuda")
        line = f"{M}x{N}x{K}:"
        for skinny in (True, False):
            ops.SKINNY = skinny
            f = timeit(lambda: ops.linear_fwd(x, w, b))
            dw = timeit(lambda: ops.linear_bwd_weight(dy, x))
            dx = timeit(lambda: ops.linear_bwd_input(dy, w))
            line += f"  {'skinny' if skinny else 'gemm  '} fwd {f:6.1f} dW {dw:6.1f} dX {dx:6.1f} us |"
        print(line)
