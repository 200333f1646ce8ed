import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
from mapx import ops
from gemm_bench import timeit
dev = "cuda"
for name, M, N, K, akc, bkc in (("dw t2s1", 1000, 1000, 4096, False, False), ("fwd t2", 4096, 1000, 1000, True, True),
                                ("fwd t0", 4096, 1000, 1000, True, True)):
    a = torch.randn(M, K, device=dev) if akc else torch.randn(K, M, device=dev)
    b = torch.randn(N, K, device=dev) if bkc else torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev)
    tile = 0 if name.endswith("t0") else 2
    res = []
    for dbg in (0, 1, 2, 3):
        us = timeit(lambda: ops.gemm(a, b, akc, bkc, M, N, K, out=out, tile=tile | (dbg << 8)))
        res.append(f"dbg{dbg}: {us:6.1f} us")
    print(name, " | ".join(res), f"| ideal@2.37GHz {2.0*M*N*K/155e6:6.1f} us (all CUs)")
