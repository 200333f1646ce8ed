"""Time of the bf16 GEMM against K (fixed cost vs slope) and for the step's shapes (tools; GPU box).
    python tools/gemm_bf16_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
from mapx import ops  # noqa: E402
from mapx.native import EPI_BIAS_RELU, EPI_NONE  # noqa: E402

BF = torch.bfloat16


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(40_000_000)        # park the GPU (~20 ms): the host enqueues all launches ahead of it
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def run(a_kc, b_kc, M, N, K, epi=EPI_NONE, out_dtype=BF, nsplit=1, tile=-1):
    A = torch.randn((M, K) if a_kc else (K, M), device="cuda").to(BF)
    B = torch.randn((N, K) if b_kc else (K, N), device="cuda").to(BF)
    bias = torch.randn(N, device="cuda") if epi != EPI_NONE else None
    out = torch.empty(M, N, dtype=out_dtype, device="cuda")
    us = timeit(lambda: ops.gemm_bf16(A, B, a_kc, b_kc, M, N, K, out=out, epi=epi, bias=bias, nsplit=nsplit, tile=tile))
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__":
    print("NT 4096 x 1000 x K, bias+relu, bf16 out")
    for K in (64, 128, 256, 512, 1024, 2048, 4096):
        us, tf = run(True, True, 4096, 1000, K, epi=EPI_BIAS_RELU)
        print(f"  K={K:5d}: {us:7.1f} us  {tf:7.1f} TF")
    print("ablation at 4096 x 1000 (tile 128x128): full | no global stores | no K loop | neither")
    for K in (64, 1024):
        r = [run(True, True, 4096, 1000, K, epi=EPI_BIAS_RELU, tile=2 + (d << 8))[0] for d in (0, 1, 2, 3)]
        print(f"  K={K:5d}: " + " | ".join(f"{x:6.1f}" for x in r) + " us")
    if len(sys.argv) > 1 and sys.argv[1] == "ablate":
        sys.exit(0)
    print("step shapes")
    for name, args in [("fwd 4096x1000x368", (True, True, 4096, 1000, 368, EPI_BIAS_RELU)),
                       ("fwd 4096x1000x1000", (True, True, 4096, 1000, 1000, EPI_BIAS_RELU)),
                       ("fwd 4096x368x368", (True, True, 4096, 368, 368, EPI_NONE)),
                       ("fwd 4096x736x1368 f32out", (True, True, 4096, 736, 1368, EPI_NONE, torch.float32)),
                       ("dx 4096x1000x1000", (True, False, 4096, 1000, 1000)),
                       ("dx 4096x1368x736", (True, False, 4096, 1368, 736)),
                       ("dx 4096x368x1000", (True, False, 4096, 368, 1000)),
                       ("dw 1000x1000x4096 ns4", (False, False, 1000, 1000, 4096, EPI_NONE, torch.float32, 4)),
                       ("dw 1000x1000x4096 ns1 t0", (False, False, 1000, 1000, 4096, EPI_NONE, torch.float32, 1, 0)),
                       ("dw 1000x368x4096 ns8", (False, False, 1000, 368, 4096, EPI_NONE, torch.float32, 8)),
                       ("dw 736x1368x4096 ns4", (False, False, 736, 1368, 4096, EPI_NONE, torch.float32, 4)),
                       ("dw 368x368x4096 ns16", (False, False, 368, 368, 4096, EPI_NONE, torch.float32, 16))]:
        for tile in ((-1,) if len(args) > 8 else (-1, 1)):
            a = list(args) + [EPI_NONE, BF, 1][len(args) - 5:] if len(args) < 8 else list(args[:8])
            us, tf = run(a[0], a[1], a[2], a[3], a[4], epi=a[5], out_dtype=a[6], nsplit=a[7],
                         tile=args[8] if len(args) > 8 else tile)
            print(f"  {name:28s} tile {tile:2d}: {us:7.1f} us  {tf:7.1f} TF")
