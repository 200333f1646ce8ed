"""Step time of the OTHER BASELINE configs on one GPU (no bench line: the headline metric is bench.py's MFP step):
configs[3] DCNv2 + RFD (Unigram replacement) and configs[4] DCNv2 finetune (CTR), Avazu / Criteo shapes, batch 4096,
the Trainer's own captured step.
    python tools/step_bench.py --pt RFD|CTR|MFP [--workload avazu|criteo] [--dtype f32|bf16] [--steps 200]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts map-code_amd on the path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pt", default="RFD", choices=["MFP", "RFD", "CTR"])
    ap.add_argument("--model", default="DCNv2", choices=["DCNv2", "DNN", "DeepFM", "xDeepFM", "AutoInt"])
    ap.add_argument("--workload", default="avazu", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--rows", type=int, default=1 << 22)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preroll", type=int, default=200)
    ap.add_argument("--tensors", action="store_true", help="deal the batches as tensors (A/B of the row references)")
    a = ap.parse_args()
    a.uniform = False
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    tr, cfg, ids, labels, _ = bench.build(a, device, 0)
    train = tr._begin("bench")
    gen = tr._generator()
    kind = a.pt.lower()
    rows = not a.tensors                       # as the Trainer's loops deal them: row references into the split
    state = {"it": train.batches(a.batch, True, gen, (0, 1), rows=rows)}

    def next_batch():
        try:
            return next(state["it"])
        except StopIteration:
            state["it"] = train.batches(a.batch, True, gen, (0, 1), rows=rows)
            return next(state["it"])

    tr.model.train()
    for _ in range(a.preroll + a.warmup):
        tr.run_step(kind, *next_batch())
    live = [g for g in tr._graphs.values() if not isinstance(g, int)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = tr.run_step(kind, *next_batch())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"step": f"{a.model} {a.pt}", "workload": a.workload, "dtype": a.dtype, "batch": a.batch,
                      "ms_per_step": 1e3 * dt / a.steps, "samples_per_s": a.batch * a.steps / dt,
                      "graphed": bool(tr.use_graph and live), "loss": float(out[0].detach())}))


if __name__ == "__main__":
    main()
