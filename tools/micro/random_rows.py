"""What random 128-byte rows cost on this GPU, as a yardstick for the table kernels (table_adam_*,
nce_table_grad): a [V, 32] fp32 table of Avazu's size, n distinct random rows (n = the rows one step's
NCE samples touch), moved by the simplest possible kernels (torch.index_select = random read + streamed
write; index_copy_ = streamed read + random write).  The lazy-AdamW update does 3 such reads + 3 such
writes per row (p, m, v) plus the 4-byte `last` entry; its time is printed beside their sum.
    python tools/micro/random_rows.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "map-code_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gemm_f32_bench import timeit  # noqa: E402

if __name__ == "__main__":
    V, W = 9_449_445, 32
    dev = "cuda"
    tabs = [torch.randn(V, W, device=dev) for _ in range(3)]
    for n in (17_296, 85_977):
        g = torch.Generator(device=dev).manual_seed(n)
        # the step's rows are skewed towards small ids (frequent values first in every field): mimic with a
        # mix of a dense head and a uniform tail, then make them distinct and sorted as the plan does
        idx = torch.cat([torch.randint(0, 200_000, (n // 2,), device=dev, generator=g),
                         torch.randint(0, V, (n,), device=dev, generator=g)]).unique()[:n].contiguous()
        n = idx.numel()
        rows = torch.randn(n, W, device=dev)
        rd = timeit(lambda: torch.index_select(tabs[0], 0, idx))
        wr = timeit(lambda: tabs[1].index_copy_(0, idx, rows))
        b = n * W * 4
        print(f"n = {n}: random read  {rd:6.1f} us = {2 * b / rd / 1e3:6.0f} GB/s (random + streamed bytes);  "
              f"random write {wr:6.1f} us = {2 * b / wr / 1e3:6.0f} GB/s")
        print(f"          3 reads + 3 writes of that kind: {3 * (rd + wr):6.1f} us for {6 * b / 1e6:.0f} MB of rows "
              f"-> {6 * b / (3 * (rd + wr)) / 1e3:.0f} GB/s of random-row traffic")
