#!/usr/bin/env python3
"""Print the kernel timeline of one training step from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace -d DIR -o NAME -- python3 bench.py ...` writes DIR/NAME_results.db).
    python tools/step_timeline.py <results.db> [step_index] [marker]
A step runs from one launch of the marker kernel (default mask_mfp_kernel) to the next."""
import re
import sqlite3
import sys

db, which = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
c = sqlite3.connect(db)
rows = c.execute("select name,start,end,queue_id,grid_x,workgroup_x from kernels order by start").fetchall()
marker = sys.argv[3] if len(sys.argv) > 3 else "mask_mfp_kernel"
marks = [i for i, r in enumerate(rows) if marker in r[0]]
lo, hi = marks[which], marks[which + 1]
t0 = rows[lo][1]
for name, s, e, q, g, w in rows[lo:hi]:
    nm = re.sub(r"\(.*", "", name).replace("void ", "").replace("mapx::", "")
    nm = re.sub(r"rocprim::ROCPRIM_\w+::detail::", "rp::", nm)[:60]
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  q{q} g{g // max(w, 1):5d}  {nm}")
print("step span", (rows[hi][1] - t0) / 1e3, "us")
