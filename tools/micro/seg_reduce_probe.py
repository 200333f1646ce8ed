"""The NCE table's gradient reduction (csrc/segreduce.h: seg_reduce_pass_a / _b over NceContrib) on ONE REAL step's
plan, alone, beside tools/micro/seg_reduce_floor.hip's stripped chain on the same plan.
Runs a few eager steps of bench.py's workload, captures the arguments of the step's ops.nce_table_grad call, times
that call, dumps the plan's (perm, rank) and runs the floor program on it.
    python tools/micro/seg_reduce_probe.py [--rows N]"""
import argparse
import os
import subprocess
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "map-code_amd"))
import bench  # noqa: E402
from mapx import ops  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(20_000_000)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1 << 20)
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--rows", str(a.rows), "--no-cpu-baseline"]
    args = bench.parse()
    dev = torch.device("cuda:0")
    tr, cfg, ids, labels, _ = bench.build(args, dev, 0)
    tr.use_graph = False
    got = {}
    real = ops.nce_table_grad

    def spy(plan, dlogit, h, K, P, gscale=None):
        got["args"] = (plan, dlogit.clone(), h.clone(), K, P, None if gscale is None else gscale.clone())
        return real(plan, dlogit, h, K, P, gscale)

    ops.nce_table_grad = spy
    train = tr._begin("probe")
    tr.model.train()
    n = 0
    for X, Y in train.batches(args.batch, True, tr._generator(), (0, 1)):
        tr.run_step("mfp", X, Y)
        n += 1
        if n == 30:
            break
    ops.nce_table_grad = real
    plan, dlogit, h, K, P, gs = got["args"]
    torch.cuda.synchronize()
    U = int(plan.n_uniq[0].item()) if plan.n_uniq is not None else -1
    print(f"plan: n = {plan.n} entries, {U} distinct rows; T = {h.shape[0]}, K + 1 = {K + 1}, P = {P}")

    def call():
        return real(plan, dlogit, h, K, P, gs)

    us = timeit(call)
    print(f"mapx_nce_table_grad (pass A + pass B + the counter's memset): {us:.1f} us")
    perm, rank = plan.perm[:plan.n].cpu().numpy().astype("int32"), plan.rank[:plan.n].cpu().numpy().astype("int32")
    path = "/tmp/seg_plan.bin"
    with open(path, "wb") as f:
        f.write(perm.tobytes())
        f.write(rank.tobytes())
    exe = "/tmp/srf"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "micro", "seg_reduce_floor.hip"),
                    "-o", exe], check=True)
    assert plan.n == h.shape[0] * (K + 1)
    print(subprocess.run([exe, "0", path, str(h.shape[0]), str(K + 1)], capture_output=True, text=True).stdout)
