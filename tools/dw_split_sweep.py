"""Split-K / tile sweep of the weight-gradient products (dW = dY^T X, TN) on the step's shapes, including
the slab reduce.   MAPX_GEMM=x3 python tools/dw_split_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mapx import ops  # noqa: E402
from gemm_f32_bench import timeit  # noqa: E402

if __name__ == "__main__":
    for (M, N, K) in [(368, 368, 4096), (1000, 368, 4096), (1000, 1000, 4096), (736, 1368, 4096), (32, 1368, 4096)]:
        A = torch.randn(K, M, device="cuda")
        B = torch.randn(K, N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        best = None
        for tile in (0, 1, 3):
            row = []
            for ns in (1, 2, 4, 8, 16):
                us = timeit(lambda: ops.gemm(A, B, False, False, M, N, K, out=out, nsplit=ns, tile=tile))
                row.append(f"ns{ns}: {us:6.1f}")
                if best is None or us < best[0]:
                    best = (us, tile, ns)
            print(f"dw {M}x{N}x{K} tile {tile}: " + "  ".join(row))
        print(f"   best {best[0]:.1f} us (tile {best[1]}, ns {best[2]}) = {2.0 * M * N * K / best[0] / 1e6:.1f} TF;  "
              f"default: {timeit(lambda: ops.linear_bwd_weight(A.t().contiguous().t() if False else torch.randn(K, M, device='cuda'), torch.randn(K, N, device='cuda'))):.1f} us")
