#!/usr/bin/env python3
"""Fixed (K-independent) cost of one GEMM launch: time vs K at the step's M, N (GPU box).
    python tools/gemm_fixed_cost.py
Prints rocprof-free event timings of back-to-back launches; the K -> 0 intercept is
launch + prologue + epilogue, the slope is the main loop."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
from mapx import ops  # noqa: E402
from mapx import native as N  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def graphed(fn, reps=20):
    """GPU-side time per launch: `reps` dependent launches replayed from a hipGraph (eager
    back-to-back launches of short kernels measure the host, not the GPU)."""
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    return timeit(g.replay, 20) / reps


M, Nn = 4096, 1000
for tile in (2, 0):
    for K in (32, 128, 512, 1000, 2000):
        a, b = torch.randn(M, K, device="cuda"), torch.randn(Nn, K, device="cuda")
        bias = torch.randn(Nn, device="cuda")
        out = torch.empty(M, Nn, device="cuda")
        add = torch.randn(M, Nn, device="cuda")
        t_plain = graphed(lambda: ops.gemm(a, b, True, True, M, Nn, K, out=out, tile=tile))
        t_epi = graphed(lambda: ops.gemm(a, b, True, True, M, Nn, K, out=out, tile=tile, epi=N.EPI_BIAS_RELU, bias=bias))
        t_add = graphed(lambda: ops.gemm(a, b, True, True, M, Nn, K, out=out, tile=tile, epi=N.EPI_ADD, aux1=add))
        print(f"tile {tile} K {K:5d}: plain {t_plain:6.1f} us   bias+relu {t_epi:6.1f} us   add {t_add:6.1f} us")
