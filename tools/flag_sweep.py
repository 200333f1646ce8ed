#!/usr/bin/env python3
"""Greedy sweep over the step's scheduling switches on ONE box (they were tuned one at a time against a moving
baseline, and their effects do not add: the graph runtime's queue assignment changes with every one of them).
Starts from the defaults, flips one switch at a time, keeps a flip that wins by > 0.3 % (mean of two runs),
two passes.      gpurun -- 'python3 tools/flag_sweep.py [--dtype bf16] > gpurun_out/flag_sweep.log'"""
import json
import os
import subprocess
import sys

FLAGS = [("MAPX_JOIN_FUSE", "1", "0"), ("MAPX_CROSS_FUSE", "0", "1"), ("MAPX_DW_BATCH", "1", "0"),
         ("MAPX_X0_LINK", "1", "0"), ("MAPX_LAYOUT_ON_MAIN", "auto", "1"), ("MAPX_PLAN_AFTER_DNN", "1", "0"),
         ("MAPX_HEAD_SIDE", "1", "0"), ("MAPX_LATE_TABLE", "1", "0"), ("MAPX_TAIL_OVERLAP", "1", "0"),
         ("MAPX_NCE_EARLY", "1", "0"), ("MAPX_JOINT_PLAN", "auto", "off"), ("MAPX_RELU_LINK", "1", "0"),
         ("MAPX_EARLY_TABLE_UPDATE", "0", "1"), ("MAPX_XCD_SLICES", "1", "0"), ("MAPX_PACK_MOMENTS", "1", "0"),
         ("MAPX_PLAN_IMPLIED", "1", "0"), ("MAPX_HEAD_SIDE_DP", "1", "0"), ("MAPX_WALK", "1", "0"),
         # round 4's switches
         ("MAPX_JOIN_DEEP_FIRST", "1", "0"), ("MAPX_PLANES_AT_START", "0", "1"), ("MAPX_CATCHUP_AFTER_CROSS", "0", "1"),
         ("MAPX_DX_FIRST", "0", "1"), ("MAPX_LATE_DENSE_FIRST", "0", "1"), ("MAPX_HEAD_DW_LATE", "1", "0"),
         ("MAPX_TOTALS_LATER", "1", "0"), ("MAPX_DEFER_COLSUM", "1", "0"), ("MAPX_TABLE_ON_PLAN", "auto", "0")]
extra = sys.argv[1:]
root = os.environ.get("GRAFT_REPO_ROOT", ".")


def run(env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"] + extra, env=e,
                         capture_output=True, text=True)
    try:
        return json.loads(out.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        print("  run failed:", out.stderr[-300:], flush=True)
        return float("inf")


def mean2(env):
    a, b = run(env), run(env)
    return (a + b) / 2, (a, b)


cur = {k: d for k, d, _ in FLAGS}
best, runs = mean2(cur)
print(f"defaults: {best:.4f} {runs}", flush=True)
for sweep in range(1, int(os.environ.get('SWEEP_PASSES', '2')) + 1):
    changed = False
    for k, d, alt in FLAGS:
        trial = dict(cur)
        trial[k] = alt if cur[k] == d else d
        m, runs = mean2(trial)
        keep = m < best * 0.997
        print(f"pass {sweep}: {k}={trial[k]}: {m:.4f} {runs}  (best {best:.4f}) {'KEEP' if keep else ''}", flush=True)
        if keep:
            cur, best, changed = trial, m, True
    if not changed:
        break
print("final:", {k: v for k, v in cur.items() if v != dict((a, b) for a, b, _ in FLAGS)[k]}, f"{best:.4f}", flush=True)
