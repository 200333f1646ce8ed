"""Max error of ops.gemm against an fp64 product for a list of (layout, M, N, K) shapes, relative to
the forward-error scale sum_k |a||b| (tools; run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
from mapx import ops  # noqa: E402


def check(a_kc, b_kc, M, N, K, tile=-1, nsplit=1):
    g = torch.Generator().manual_seed(M * 131 + N * 17 + K)
    A = torch.randn((M, K) if a_kc else (K, M), generator=g)
    B = torch.randn((N, K) if b_kc else (K, N), generator=g)
    Am = A if a_kc else A.t()
    Bm = B.t() if b_kc else B
    ref = Am.double() @ Bm.double()
    bound = Am.abs().double() @ Bm.abs().double()
    out = ops.gemm(A.cuda(), B.cuda(), a_kc, b_kc, M, N, K, tile=tile, nsplit=nsplit)
    err = ((out.double().cpu() - ref).abs() / bound.clamp(min=1e-30)).max().item()
    return err


if __name__ == "__main__":
    bad = 0
    for (a_kc, b_kc, M, N, K) in [(1, 0, 777, 1248, 39), (1, 0, 64, 1248, 39), (1, 0, 777, 1248, 23), (1, 0, 777, 1248, 40),
                                  (1, 0, 4096, 736, 23), (1, 0, 777, 1248, 33), (1, 0, 9, 1248, 39), (1, 0, 137, 100, 39),
                                  (1, 1, 777, 39, 1248), (0, 0, 39, 1248, 777), (0, 0, 1248, 1624, 777), (1, 0, 777, 1624, 1248),
                                  (1, 1, 4096, 1000, 368), (1, 1, 4096, 1000, 1000), (1, 0, 4096, 1000, 1000),
                                  (1, 0, 4096, 1368, 736), (0, 0, 1000, 1000, 4096), (1, 1, 256, 128, 512),
                                  (1, 1, 300, 200, 200), (0, 0, 300, 200, 520)]:
        for tile in (-1, 0, 1, 2, 3):
            e = check(bool(a_kc), bool(b_kc), M, N, K, tile)
            flag = "" if e < 1e-6 else "   <-- BAD"
            bad += e >= 1e-6
            print(f"a_kc={a_kc} b_kc={b_kc} M={M} N={N} K={K} tile={tile}: err/bound {e:.2e}{flag}")
    print("bad:", bad)
