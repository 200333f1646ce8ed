"""Which tile layout the fp32 GEMM should take for a shape (0: 64 x 64, 1: 128 x 64, 2: 128 x 128 with 8 waves,
3: 128 x 128 woven):   python tools/micro/tile_probe.py M N K [M N K ...]"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "map-code_amd"))
sys.path.insert(0, os.path.join(HERE, ".."))
from mapx import ops  # noqa: E402
from gemm_f32_bench import timeit  # noqa: E402

if __name__ == "__main__":
    v = [int(a) for a in sys.argv[1:]] or [4096, 624, 624, 4096, 368, 368, 4096, 1000, 624]
    for M, N, K in zip(v[0::3], v[1::3], v[2::3]):
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        wt = torch.randn(K, N, device="cuda")
        dy = torch.randn(M, K, device="cuda")
        line = f"{M}x{N}x{K}:"
        for name, fn in (("fwd", lambda t: ops.gemm(x, w, True, True, M, N, K, tile=t)),
                         ("dX ", lambda t: ops.gemm(dy, wt, True, False, M, N, K, tile=t))):
            line += f"  {name}"
            for t in (-1, 0, 1, 3):
                us = timeit(lambda: fn(t))
                line += f" t{t} {us:5.1f}"
        print(line)
