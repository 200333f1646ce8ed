#!/usr/bin/env python3
"""Matrix-pipe busy share per kernel from one rocprofv3 --pmc pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \\
              -d DIR -o m -- python3 bench.py ...            (MAPX_GRAPH=0 MAPX_SERIAL=1: every kernel alone)
    python tools/pmc_mfma_summary.py <counter_collection.csv> [out.json]
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs (32 per v_mfma_f32_32x32x16_*), GRBM_GUI_ACTIVE the
dispatch's cycles summed over the 8 XCDs (MI355X_MICROARCH.md): busy share = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs)."""
import csv
import json
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[name] += 1
out = {}
for k, d in acc.items():
    if not k.startswith("mapx::") or not d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        continue
    simd_cycles = d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    out[k] = dict(launches=cnt[k], mfma_busy_cycles=d["SQ_VALU_MFMA_BUSY_CYCLES"], simd_cycles=simd_cycles,
                  mfma_busy=d["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                  gpu_cycles_per_launch=d["GRBM_GUI_ACTIVE"] / 8.0 / max(cnt[k], 1))
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles"]):
    print(f"{k[:76]:76s} x{v['launches']:4d}  matrix pipe busy {100 * v['mfma_busy']:5.1f} %   {v['gpu_cycles_per_launch']:9.0f} cycles / launch")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
