#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM on the shapes of one DCNv2+MFP step (GPU box).
    python tools/gemm_bench.py            # every shape x tile x split, best per shape
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
from mapx import ops  # noqa: E402

B = 4096
D, H, FP = 368, 1000, 736
FWD = [("cross fwd", B, D, D), ("dnn0 fwd", B, H, D), ("dnn1 fwd", B, H, H), ("enc fwd", B, FP, D + H)]
DX = [("cross dx", B, D, D), ("dnn0 dx", B, D, H), ("dnn1 dx", B, H, H), ("enc dx", B, D + H, FP)]
DW = [("cross dw", D, D, B), ("dnn0 dw", H, D, B), ("dnn1 dw", H, H, B), ("enc dw", FP, D + H, B)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    dev = "cuda"
    for kind, shapes in (("fwd", FWD), ("dx", DX), ("dw", DW)):
        for name, M, N, K in shapes:
            if kind == "fwd":
                a, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
                akc, bkc = True, True
            elif kind == "dx":
                a, b = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
                akc, bkc = True, False
            else:
                a, b = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
                akc, bkc = False, False
            out = torch.empty(M, N, device=dev)
            res = []
            for tile in (2, 3, 0, 4):
                for ns in ((1,) if kind != "dw" else (1, 2, 4, 8, 16)):
                    us = timeit(lambda: ops.gemm(a, b, akc, bkc, M, N, K, out=out, nsplit=ns, tile=tile))
                    res.append((us, tile, ns))
            ref = timeit(lambda: torch.mm(a if akc else a.t(), b.t() if bkc else b))
            fl = 2.0 * M * N * K
            best = min(res)
            line = "  ".join(f"t{t}s{ns}:{us:6.1f}" for us, t, ns in res)
            print(f"{name:10s} M{M:5d} N{N:5d} K{K:5d}  best {best[0]:6.1f} us ({fl / best[0] / 1e6:5.1f} TF, tile {best[1]} split {best[2]})"
                  f"  torch.mm {ref:6.1f} us ({fl / ref / 1e6:5.1f} TF) | {line}")


if __name__ == "__main__":
    main()
