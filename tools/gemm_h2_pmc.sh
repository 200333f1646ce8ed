#!/bin/bash
# SQ / TCP counters of the fp32 GEMM kernels on one shape (default: NT 4096 x 1000 x 4096), one --pmc pass per counter
# group (rocprofv3; program directly after --):   gpurun -- 'bash tools/gemm_h2_pmc.sh [h2|x3] [tile]'
set -e
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
MODE=${1:-h2}; TILE=${2:-3}
O=gpurun_out/h2_pmc_${MODE}_t${TILE}
rm -rf $O && mkdir -p $O
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p -- python3 tools/gemm_h2_bench.py once $MODE $TILE > $O/p$i.log 2>&1 || { tail -5 $O/p$i.log; continue; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {sum(v) / len(v):16.0f}   ({len(v)} launches)")
PY
rm -rf $O/p*/
