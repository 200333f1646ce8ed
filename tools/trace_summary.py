#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel time per step over the LAST `steps`
steps of a bench run, plus GPU idle time between kernels.
    python tools/trace_summary.py <kernel_trace.csv> <steps> [kernels_per_step_marker]
"""
import csv
import re
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step = from one mask_mfp_kernel launch to the next
marks = [i for i, r in enumerate(rows) if "mask_mfp_kernel" in r["Kernel_Name"]]
lo = marks[-steps - 1] if len(marks) > steps else marks[0]
hi = marks[-1]
sel = rows[lo:hi]
nsteps = len([m for m in marks if lo <= m < hi])
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
busy = 0
per = defaultdict(lambda: [0, 0])
prev_end = t0
gaps = 0
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    name = re.sub(r"rocprim::ROCPRIM_\w+::detail::", "rocprim::", name)[:70]
    per[name][0] += e - s
    per[name][1] += 1
    busy += e - s
    if s > prev_end:
        gaps += s - prev_end
    prev_end = max(prev_end, e)
wall = t1 - t0
print(f"steps {nsteps}  wall/step {wall / nsteps / 1e3:.1f} us  kernel-busy/step {busy / nsteps / 1e3:.1f} us  "
      f"idle gaps/step {gaps / nsteps / 1e3:.1f} us  launches/step {len(sel) / nsteps:.0f}")
for name, (ns, n) in sorted(per.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{ns / nsteps / 1e3:8.1f} us/step  x{n / nsteps:5.1f}  avg {ns / n / 1e3:7.1f} us  {name}")
