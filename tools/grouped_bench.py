"""GPU time of the grouped feat_encoder products on the step's shape (B 4096, F 23, L 6, D+H 1368).
MAPX_GEMM=mfma32|x3 selects the kernel family.  With `once`: a few launches only (for rocprofv3 --pmc).
    MAPX_GEMM=x3 python tools/grouped_bench.py [once]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "map-code_amd"))
from mapx import ops  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_f32_bench import timeit  # noqa: E402

if __name__ == "__main__":
    B, F, L, K, P = 4096, 23, 6, 1368, 32
    g = torch.Generator().manual_seed(1)
    mi = torch.stack([torch.randperm(F, generator=g)[:L] for _ in range(B)]).cuda()
    final = torch.randn(B, K, device="cuda")
    w = torch.randn(F * P, K, device="cuda") / K ** 0.5
    b = torch.randn(F * P, device="cuda")
    groups = ops.EncGroups(mi, F)
    dh = torch.randn(groups.cap, P, device="cuda")
    out = torch.empty(F * P, K, device="cuda")
    fwd = lambda: ops.enc_grouped_fwd(final, w, b, groups)
    dw = lambda: ops.enc_grouped_dw(dh, final, groups, out=out)
    if len(sys.argv) > 1 and sys.argv[1] == "once":
        for _ in range(3):
            fwd(); dw()
        torch.cuda.synchronize()
        sys.exit(0)
    flop = 2.0 * B * L * P * K
    print("mode", os.environ.get("MAPX_GEMM", "x3"), "slots", groups.cap, "used tiles", int(groups.group_start[-1]) // 128)
    for name, fn in (("fwd", fwd), ("dw", dw)):
        us = timeit(fn)
        print(f"  grouped {name:4s}: {us:7.1f} us  {flop / us / 1e6:6.1f} TF")
