#!/bin/bash
# HBM-side traffic of the grouped feat_encoder kernels:  gpurun -- 'bash tools/grouped_pmc.sh'
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/grouped_pmc
rm -rf $O && mkdir -p $O
for M in x3 mfma32; do
  MAPX_GEMM=$M python3 tools/grouped_bench.py
  MAPX_GEMM=$M rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$M -o f -- python3 tools/grouped_bench.py once > $O/f_$M.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob("$O/f_$M/**/*counter_collection.csv", recursive=True)[0])):
    if "grouped" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]) * 1024 * 2 / 1e6)
for k, v in acc.items():
    print(f"   $M {k}: fetch(x2) {sum(v) / len(v):8.1f} MB per launch")
PY
done
rm -rf $O/f_*/
