"""Does a 256-byte record (m | v of a row side by side) move faster than two 128-byte rows at unrelated addresses?
The question behind packing the lazy-AdamW moments of a table row into one line pair.
Random distinct rows of a [9.4 M, 32] fp32 table (x2) against the same rows of one [9.4 M, 64] table, read
(index_select) and written (index_copy_).      python tools/micro/row_width.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "map-code_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gemm_f32_bench import timeit  # noqa: E402

if __name__ == "__main__":
    V, dev = 9_449_445, "cuda"
    a, b = torch.randn(V, 32, device=dev), torch.randn(V, 32, device=dev)
    ab = torch.randn(V, 64, device=dev)
    for n in (17_296, 85_977):
        g = torch.Generator(device=dev).manual_seed(n)
        idx = torch.cat([torch.randint(0, 200_000, (n // 2,), device=dev, generator=g),
                         torch.randint(0, V, (n,), device=dev, generator=g)]).unique()[:n].contiguous()
        n = idx.numel()
        r32, r64 = torch.randn(n, 32, device=dev), torch.randn(n, 64, device=dev)
        rd2 = timeit(lambda: (torch.index_select(a, 0, idx), torch.index_select(b, 0, idx)))
        rd1 = timeit(lambda: torch.index_select(ab, 0, idx))
        wr2 = timeit(lambda: (a.index_copy_(0, idx, r32), b.index_copy_(0, idx, r32)))
        wr1 = timeit(lambda: ab.index_copy_(0, idx, r64))
        print(f"n = {n}: read  2 x 128 B {rd2:6.1f} us   1 x 256 B {rd1:6.1f} us | write 2 x 128 B {wr2:6.1f} us   "
              f"1 x 256 B {wr1:6.1f} us   (two launches vs one: subtract ~2 us)")
