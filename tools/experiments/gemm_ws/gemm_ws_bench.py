"""Wave-specialised fp32 GEMM (csrc/gemm_ws.hip) against the woven kernel of gemm_x3.hip: error against fp64
and GPU time on the step's big shapes, both operand sources (mode 0: bf16 planes by LDS-DMA, mode 1: fp32
cut by the loader waves), with and without raised consumer priority.
    python tools/experiments/gemm_ws/gemm_ws_bench.py [check|time|all]"""
import os
import sys

import torch

import ctypes as C

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "..", "map-code_amd"))
from mapx import ops  # noqa: E402
from mapx.native import EPI_BIAS_RELU, EPI_NONE, check, stream  # noqa: E402

lib = C.CDLL(os.path.join(HERE, "libgemm_ws.so"))        # make -C tools/experiments/gemm_ws
_p, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
lib.mapx_cut_planes.restype, lib.mapx_cut_planes.argtypes = _i, [_p, _i64, _i, _i, _p, _i64, _i64, _p]
lib.mapx_gemm_ws.restype = _i
lib.mapx_gemm_ws.argtypes = [_i, _i, _i, _i, _i, _i, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i, _p, _p, _i64, _p, _i64,
                             _p, _i64, _i, _p, _sz, _p]

BF16 = torch.bfloat16


def planes_of(x):
    """fp32 [R, C] -> bf16 [3, R, C] (hi, mid, lo) by the library's cut."""
    R, C = x.shape
    p = torch.empty(3, R, C, dtype=BF16, device=x.device)
    check(lib.mapx_cut_planes(x.data_ptr(), x.stride(0), R, C, p.data_ptr(), C, R * C, stream()))
    return p


def gemm_ws(mode, A, B, a_kc, b_kc, M, N, K, out, epi=EPI_NONE, bias=None, nsplit=1, ws=None):
    pl = (mode & 25) == 0
    lda = A.shape[-1]
    ldb = B.shape[-1]
    pa = A.shape[-2] * A.shape[-1] if pl else 0
    pb = B.shape[-2] * B.shape[-1] if pl or (mode & 16) else 0      # 16: hybrid, A fp32 / B planes
    check(lib.mapx_gemm_ws(mode, int(a_kc), int(b_kc), M, N, K, A.data_ptr(), lda, pa, B.data_ptr(), ldb, pb,
                           out.data_ptr(), out.stride(0), epi, bias.data_ptr() if bias is not None else None,
                           None, 0, None, 0, None, 0, nsplit, ws.data_ptr() if ws is not None else None,
                           ws.numel() * 4 if ws is not None else 0, stream()))
    return out


def timeit(fn, reps=20):
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


def operands(a_kc, b_kc, M, N, K, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    A = torch.randn((M, K) if a_kc else (K, M), device="cuda", generator=g)
    B = torch.randn((N, K) if b_kc else (K, N), device="cuda", generator=g)
    return A, B


def ref64(A, B, a_kc, b_kc):
    a = A.double() if a_kc else A.double().t()
    b = B.double().t() if b_kc else B.double()
    return a @ b


def check_shapes():
    worst = 0.0
    for (a_kc, b_kc, M, N, K, ns) in [(True, True, 256, 256, 64, 1), (True, True, 4096, 1000, 1000, 1),
                                      (True, True, 777, 368, 368, 1), (True, False, 4096, 1000, 1000, 1),
                                      (True, False, 1000, 1368, 736, 1), (False, False, 1000, 1000, 4096, 4),
                                      (False, False, 368, 1000, 4096, 1), (False, False, 1000, 368, 777 // 8 * 8, 2),
                                      (True, True, 130, 72, 40, 1), (True, False, 64, 8, 8, 1)]:
        A, B = operands(a_kc, b_kc, M, N, K, seed=M + N + K)
        R = ref64(A, B, a_kc, b_kc)
        scale = float(R.abs().max())
        ws = torch.empty(ns * M * N, device="cuda") if ns > 1 else None
        row = []
        for mode in (0, 1, 4, 5, 8, 16):
            out = torch.full((M, N), float("nan"), device="cuda")
            a_, b_ = (planes_of(A), planes_of(B)) if (mode & 9) == 0 else (A, B)
            if mode & 16:
                if B.shape[-1] % 8:
                    row.append(float("nan"))
                    continue
                a_, b_ = A, planes_of(B)
            gemm_ws(mode, a_, b_, a_kc, b_kc, M, N, K, out, nsplit=ns, ws=ws)
            torch.cuda.synchronize()
            err = float((out.double() - R).abs().max()) / scale
            row.append(err)
            worst = max(worst, err)
        old = ops.gemm(A, B, a_kc, b_kc, M, N, K, nsplit=ns)
        e_old = float((old.double() - R).abs().max()) / scale
        print(f"  kc {int(a_kc)}{int(b_kc)} {M}x{N}x{K} ns{ns}: err/scale ws modes {['%.2e' % e for e in row]}  x3 {e_old:.2e}")
    # epilogue pass-through: bias + ReLU
    A, B = operands(True, True, 4096, 1000, 368, seed=5)
    bias = torch.randn(1000, device="cuda")
    R = torch.relu(ref64(A, B, True, True) + bias.double())
    for mode in (0, 1, 8):
        out = torch.empty(4096, 1000, device="cuda")
        a_, b_ = (planes_of(A), planes_of(B)) if mode == 0 else (A, B)
        gemm_ws(mode, a_, b_, True, True, 4096, 1000, 368, out, epi=EPI_BIAS_RELU, bias=bias)
        err = float((out.double() - R).abs().max()) / float(R.abs().max())
        worst = max(worst, err)
        print(f"  bias+relu mode {mode}: {err:.2e}")
    print("worst", worst)
    assert worst < 2e-6, worst


def time_shapes():
    shapes = [("fwd 4096x1000x4096", True, True, 4096, 1000, 4096, 1),
              ("fwd 4096x1000x1000", True, True, 4096, 1000, 1000, 1),
              ("fwd 4096x1000x368", True, True, 4096, 1000, 368, 1),
              ("dx 4096x1000x1000", True, False, 4096, 1000, 1000, 1),
              ("dx 4096x1368x736", True, False, 4096, 1368, 736, 1),
              ("dw 1000x1000x4096 ns4", False, False, 1000, 1000, 4096, 4),
              ("dw 1000x368x4096 ns8", False, False, 1000, 368, 4096, 8)]
    for name, a_kc, b_kc, M, N, K, ns in shapes:
        A, B = operands(a_kc, b_kc, M, N, K)
        out = torch.empty(M, N, device="cuda")
        ws = torch.empty(ns * M * N, device="cuda") if ns > 1 else None
        fl = 2.0 * M * N * K
        us = timeit(lambda: ops.gemm(A, B, a_kc, b_kc, M, N, K, out=out, nsplit=ns))
        line = f"  {name:24s} x3 {us:6.1f} us {fl / us / 1e6:6.1f} TF |"
        Ap, Bp = planes_of(A), planes_of(B)
        for mode in (0, 4, 1, 5, 8, 16, 20):
            a_, b_ = (Ap, Bp) if (mode & 9) == 0 else (A, B)
            if mode & 16:
                a_, b_ = A, Bp
            us = timeit(lambda: gemm_ws(mode, a_, b_, a_kc, b_kc, M, N, K, out, nsplit=ns, ws=ws))
            line += f" m{mode} {us:6.1f} us {fl / us / 1e6:6.1f} TF |"
        print(line)


def stamp_shapes():
    """Diagnostic build (-DMAPX_WS_STAMP): cycles and clock of the phases of one launch, median over blocks."""
    import ctypes
    import numpy as np
    lib.mapx_gemm_ws_set_stamps.restype = ctypes.c_int
    lib.mapx_gemm_ws_set_stamps.argtypes = [ctypes.c_void_p]
    for name, a_kc, b_kc, M, N, K, ns in [("fwd 4096x1000x4096", True, True, 4096, 1000, 4096, 1),
                                          ("fwd 4096x1000x1000", True, True, 4096, 1000, 1000, 1),
                                          ("dx 4096x1000x1000", True, False, 4096, 1000, 1000, 1),
                                          ("dw 1000x1000x4096 ns4", False, False, 1000, 1000, 4096, 4)]:
        A, B = operands(a_kc, b_kc, M, N, K)
        out = torch.empty(M, N, device="cuda")
        ws = torch.empty(ns * M * N, device="cuda") if ns > 1 else None
        Ap, Bp = planes_of(A), planes_of(B)
        nblk = ((M + 127) // 128) * ((N + 127) // 128) * ns
        nk = -(-(K // ns) // 32)
        for mode in (0, 1, 8, 16):
            a_, b_ = (Ap, Bp) if mode == 0 else (A, Bp) if mode == 16 else (A, B)
            st = torch.zeros(nblk * 16, dtype=torch.int64, device="cuda")
            check(lib.mapx_gemm_ws_set_stamps(st.data_ptr()))
            us = timeit(lambda: gemm_ws(mode, a_, b_, a_kc, b_kc, M, N, K, out, nsplit=ns, ws=ws), reps=40)
            torch.cuda.synchronize()
            check(lib.mapx_gemm_ws_set_stamps(None))
            raw = st.cpu().numpy().reshape(nblk, 16).astype(np.float64)
            t = raw[:, :8].reshape(nblk, 4, 2)
            accs = np.median(raw[:, 8:12], axis=0) / nk
            cyc, rt = t[:, :, 0], t[:, :, 1]
            d = lambda i, j: np.median(cyc[:, j] - cyc[:, i])
            clock = np.median((cyc[:, 2] - cyc[:, 1]) / np.maximum(rt[:, 2] - rt[:, 1], 1)) * 100.0
            span = (rt.max() - rt.min()) / 100.0
            print(f"  {name:22s} mode {mode}: {us:6.1f} us | prologue {d(0, 1):7.0f} cyc  loop {d(1, 2):8.0f} cyc = "
                  f"{d(1, 2) / nk:6.1f} / K-step  epilogue {d(2, 3):6.0f} cyc | clock {clock:5.0f} MHz | "
                  f"first-to-last stamp {span:6.1f} us | per K-step: loader issue {accs[0]:5.0f} wait {accs[1]:5.0f} "
                  f"barrier {accs[2]:5.0f}, consumer barrier {accs[3]:5.0f}")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "stamp":
        stamp_shapes()
    if what in ("check", "all"):
        check_shapes()
    if what in ("time", "all"):
        time_shapes()
