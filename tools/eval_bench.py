"""Throughput of Trainer.eval() (finetune model, forward only + on-device AUC / log-loss) on one GPU:
    python tools/eval_bench.py [--rows 1048576] [--workload avazu]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="avazu", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--rows", type=int, default=1 << 20)
    ap.add_argument("--batch", type=int, default=4096)
    a = ap.parse_args()
    a.uniform, a.pt, a.preroll, a.warmup, a.steps = False, "CTR", 0, 0, 0
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tr, cfg, ids, labels, _ = bench.build(a, device, 0)
    from mapx.dataset import OurDataset
    tr.eval_dataset = OurDataset(ids, labels)
    os.makedirs(tr.args.output_dir, exist_ok=True)        # eval() saves the best model so far
    tr._begin("bench")
    tr.eval()                                   # warm-up (allocations, the split's upload)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    log = tr.eval()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nb = -(-a.rows // a.batch)
    print({"eval_rows": a.rows, "seconds": round(dt, 4), "rows_per_s": round(a.rows / dt), "ms_per_batch": round(1e3 * dt / nb, 4),
           "auc": log["eval_auc"]})


if __name__ == "__main__":
    main()
