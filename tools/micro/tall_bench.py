import os, sys, torch
sys.path.insert(0, "/root/repo/map-code_amd"); sys.path.insert(0, "/root/repo/tools")
from mapx import ops
from gemm_f32_bench import timeit
for M, N, K in [(94208, 40, 16), (94208, 40, 40)]:
    x = torch.randn(M, K, device="cuda"); dy = torch.randn(M, N, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    line = f"{M}x{N}x{K}:"
    for sk in (True, False):
        ops.SKINNY = sk
        dw = timeit(lambda: ops.linear_bwd_weight(dy, x))
        dx = timeit(lambda: ops.linear_bwd_input(dy, w))
        f = timeit(lambda: ops.linear_fwd(x, w, b))
        line += f"  {'skinny' if sk else 'gemm'} dW {dw:6.1f} dX {dx:6.1f} fwd {f:6.1f} |"
    print(line)
