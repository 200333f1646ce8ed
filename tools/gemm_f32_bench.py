"""GPU time of the fp32 GEMM entry (mapx_gemm_f32) on the step's shapes, GPU parked so that the host
runs ahead.  MAPX_GEMM=mfma32|x3 selects the kernel family (read once per process).
    MAPX_GEMM=x3 python tools/gemm_f32_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
from mapx import ops  # noqa: E402
from mapx.native import EPI_BIAS_RELU, EPI_NONE  # noqa: E402


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(40_000_000)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def run(a_kc, b_kc, M, N, K, epi=EPI_NONE, nsplit=1, tile=-1):
    A = torch.randn((M, K) if a_kc else (K, M), device="cuda")
    B = torch.randn((N, K) if b_kc else (K, N), device="cuda")
    bias = torch.randn(N, device="cuda") if epi != EPI_NONE else None
    out = torch.empty(M, N, device="cuda")
    us = timeit(lambda: ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out, epi=epi, bias=bias, nsplit=nsplit, tile=tile))
    return us, 2.0 * M * N * K / us / 1e6


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "fixed"):
    print("mode", os.environ.get("MAPX_GEMM", "x3"))
    if len(sys.argv) > 1 and sys.argv[1] == "ablate":       # one line per 128 x 128 layout; see tools/x3_ablate.sh
        for tile in (2, 3):
            us, _ = run(True, True, 4096, 1000, 4096, epi=EPI_BIAS_RELU, tile=tile)
            print(f"  tile {tile}: {us:7.1f} us   {(us - 10) / 128:5.2f} us/K-step")
        sys.exit(0)
    print("NT 4096 x 1000 x K, bias+relu")
    for K in (32, 128, 512, 1024, 2048, 4096):
        us, tf = run(True, True, 4096, 1000, K, epi=EPI_BIAS_RELU)
        print(f"  K={K:5d}: {us:7.1f} us  {tf:7.1f} TF")
    for name, args in [("fwd 4096x1000x368", (True, True, 4096, 1000, 368, EPI_BIAS_RELU)),
                       ("fwd 4096x1000x1000", (True, True, 4096, 1000, 1000, EPI_BIAS_RELU)),
                       ("fwd 4096x368x368", (True, True, 4096, 368, 368)),
                       ("dx 4096x1000x1000", (True, False, 4096, 1000, 1000)),
                       ("dx 4096x1368x736", (True, False, 4096, 1368, 736)),
                       ("dx 4096x368x1000", (True, False, 4096, 368, 1000)),
                       ("dx 4096x368x368", (True, False, 4096, 368, 368)),
                       ("dw 1000x1000x4096 ns1", (False, False, 1000, 1000, 4096, EPI_NONE, 1)),
                       ("dw 1000x1000x4096 ns4", (False, False, 1000, 1000, 4096, EPI_NONE, 4)),
                       ("dw 1000x368x4096 ns8", (False, False, 1000, 368, 4096, EPI_NONE, 8)),
                       ("dw 368x368x4096 ns16", (False, False, 368, 368, 4096, EPI_NONE, 16))]:
        a = list(args) + [EPI_NONE, 1][len(args) - 5:]
        for tile in ((-1, 0, 1, 2, 3) if os.environ.get('MAPX_GEMM', 'x3') == 'x3' else (-1,)):
            us, tf = run(a[0], a[1], a[2], a[3], a[4], epi=a[5], nsplit=a[6], tile=tile)
            print(f"  {name:26s} tile {tile:2d}: {us:7.1f} us  {tf:7.1f} TF")


def graph_time(fn, n=50, reps=10):
    """us per call with n dependent calls captured in one hipGraph (no host launch gaps)."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (reps * n)


def fixed_cost():
    """K-independent cost of a launch inside a graph: time at K = 32 .. 512 per tile layout and epilogue."""
    for (M, N) in ((4096, 1000), (4096, 368)):
        for epi in (EPI_NONE, EPI_BIAS_RELU):
            for tile in (0, 1, 3, 2):
                row = []
                for K in (32, 64, 128, 256, 512):
                    A = torch.randn(M, K, device="cuda")
                    B = torch.randn(N, K, device="cuda")
                    bias = torch.randn(N, device="cuda") if epi != EPI_NONE else None
                    out = torch.empty(M, N, device="cuda")
                    us = graph_time(lambda: ops.gemm(A, B, True, True, M, N, K, out=out, epi=epi, bias=bias, tile=tile))
                    row.append(f"K{K}: {us:5.1f}")
                print(f"{M}x{N} epi {epi} tile {tile}: " + "  ".join(row))
    x = torch.zeros(256, device="cuda")
    print(f"empty-ish kernel (x.add_(1) on 256 floats) in the same graph harness: {graph_time(lambda: x.add_(1)):.2f} us")


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "fixed":
    fixed_cost()
