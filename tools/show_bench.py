#!/usr/bin/env python3
"""Pretty-print the kernel table of a bench.py JSON line."""
import json
import sys

d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
print(f"value {d['value']:.0f} {d['unit']}  ms/step {d['ms_per_step']:.3f}  n_gpus {d['n_gpus']}")
tot = 0.0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    tot += v["ms_per_step"]
    print(f"{k:22s} {v['ms_per_step']:.3f} ms/step  avg {v['avg_us']:7.1f} us x{v['launches_per_step']:.0f}"
          f"  {v['achieved']:8.1f} {v['unit']:8s} frac {v['frac']:.3f}")
print(f"sum of timed kernels {tot:.3f} ms/step")
if "cpu_baseline" in d:
    print("cpu_baseline", d["cpu_baseline"])
