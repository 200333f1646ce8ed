"""The two-piece fp16 arithmetic of the fp32 GEMM (csrc/gemm_h2.hip) beside the six-product bf16 one
(csrc/gemm_x3.hip) on the step's shapes: GPU time of each (GPU parked so that the host runs ahead) and the worst
error of each against fp64 relative to sum_k |a_k b_k|.
    python tools/gemm_h2_bench.py [--quick]"""
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


def run(name, a_kc, b_kc, M, N, K, epi=EPI_NONE, nsplit=1, tile=-1, check=True, scale_a=1.0, scale_b=1.0):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn((M, K) if a_kc else (K, M), device="cuda", generator=g) * scale_a
    B = torch.randn((N, K) if b_kc else (K, N), device="cuda", generator=g) * scale_b
    bias = torch.randn(N, device="cuda", generator=g) if epi != EPI_NONE else None
    out = torch.empty(M, N, device="cuda")
    ra, rb = ops.amax(A), ops.amax(B)
    res = {}
    modes = [("x3", {}), ("h2", dict(amax_a=ra, amax_b=rb))]
    if a_kc and nsplit == 1 and tile < 0 and ops.planes_wanted(M, N, K):
        modes.append(("h2w", dict(amax_a=ra, amax_b=rb, b_planes=ops.h2_weight_planes(B, b_kc, rb))))
    for mode, kw in modes:
        fn = lambda: ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out, epi=epi, bias=bias, nsplit=nsplit, tile=tile, **kw)
        us = timeit(fn)
        err = float("nan")
        if check:
            o = ops.gemm(A, B, a_kc, b_kc, M, N, K, epi=EPI_NONE, nsplit=nsplit, tile=tile, **kw)
            Am, Bm = (A if a_kc else A.t()), (B.t() if b_kc else B)
            rows = slice(0, min(M, 512))
            ref = Am[rows].double() @ Bm.double()
            mag = Am[rows].abs().double() @ Bm.abs().double()
            err = float(((o[rows].double() - ref).abs() / mag).max())
        res[mode] = (us, 2.0 * M * N * K / us / 1e6, err)
    x, h = res["x3"], res["h2"]
    w = res.get("h2w")
    print(f"  {name:28s} tile {tile:2d}: x3 {x[0]:7.1f} us {x[1]:6.1f} TF err {x[2]:.2e} | h2 {h[0]:7.1f} us {h[1]:6.1f} TF err {h[2]:.2e}"
          f" | x{ x[0] / h[0]:.2f}" + (f" | h2w {w[0]:7.1f} us {w[1]:6.1f} TF err {w[2]:.2e} x{x[0] / w[0]:.2f}" if w else ""), flush=True)


if __name__ == "__main__":
    if "once" in sys.argv:            # tools/gemm_h2_pmc.sh: a few launches of one kernel
        mode, tile = sys.argv[sys.argv.index("once") + 1], int(sys.argv[sys.argv.index("once") + 2])
        M, Nn, K = 4096, 1000, 4096
        A, B = torch.randn(M, K, device="cuda"), torch.randn(Nn, K, device="cuda")
        kw = dict(amax_a=ops.amax(A), amax_b=ops.amax(B)) if mode in ("h2", "h2w") else {}
        if mode == "h2w":             # (tile must be -1; MAPX_GEMM_H2W8=0: the 4-wave kernel)
            kw["b_planes"] = ops.h2_weight_planes(B, True, kw["amax_b"])
        out = torch.empty(M, Nn, device="cuda")
        for _ in range(5):
            ops.gemm(A, B, True, True, M, Nn, K, out=out, tile=tile, **kw)
        torch.cuda.synchronize()
        sys.exit(0)
    if "ablate" in sys.argv:          # tools/h2_ablate.sh: per-K-step time from two K
        for tile in (3, 2):
            for K in (2048, 4096):
                run(f"K={K}", True, True, 4096, 1000, K, check=False, tile=tile)
        sys.exit(0)
    if "ablatew" in sys.argv:         # tools/h2_ablate.sh with H2_ABLATE_W8=1: the weight-planes kernels
        for K in (2048, 4096):
            run(f"K={K}", True, True, 4096, 1000, K, check=False)
        sys.exit(0)
    quick = "--quick" in sys.argv
    print("NT 4096 x 1000 x K, bias+relu")
    for K in ((128, 1024, 4096) if quick else (64, 128, 512, 1024, 2048, 4096)):
        run(f"K={K}", True, True, 4096, 1000, K, epi=EPI_BIAS_RELU, check=K <= 1024)
    shapes = [("fwd 4096x1000x368", (True, True, 4096, 1000, 368, EPI_BIAS_RELU)),
              ("fwd 4096x1000x1000", (True, True, 4096, 1000, 1000, EPI_BIAS_RELU)),
              ("fwd 4096x368x368", (True, True, 4096, 368, 368)),
              ("fwd 4096x736x1368", (True, True, 4096, 736, 1368)),
              ("dx 4096x1000x1000", (True, False, 4096, 1000, 1000)),
              ("dx 4096x1368x736", (True, False, 4096, 1368, 736)),
              ("dx 4096x368x1000", (True, False, 4096, 368, 1000)),
              ("dx 4096x368x368", (True, False, 4096, 368, 368)),
              ("dw 1000x1000x4096 ns1", (False, False, 1000, 1000, 4096, EPI_NONE, 1)),
              ("dw 1000x1000x4096 ns4", (False, False, 1000, 1000, 4096, EPI_NONE, 4)),
              ("dw 1000x368x4096 ns8", (False, False, 1000, 368, 4096, EPI_NONE, 8)),
              ("dw 368x368x4096 ns16", (False, False, 368, 368, 4096, EPI_NONE, 16)),
              ("dw 736x1368x4096 ns2", (False, False, 736, 1368, 4096, EPI_NONE, 2))]
    for name, args in shapes:
        a = list(args) + [EPI_NONE, 1][len(args) - 5:]
        for tile in ((-1,) if quick else (-1, 0, 1, 3)):
            run(name, a[0], a[1], a[2], a[3], a[4], epi=a[5], nsplit=a[6], tile=tile)
    print("magnitudes: gradients ~1e-6, activations ~1e2")
    run("dx 4096x1000x1000 tiny dy", True, False, 4096, 1000, 1000, scale_a=1e-6, scale_b=0.03)
    run("fwd 4096x1000x1000 big x", True, True, 4096, 1000, 1000, scale_a=300.0, scale_b=0.03)
