#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for PT in RFD CTR; do
  for E in "MAPX_X=0" "MAPX_PLAN_AFTER_TRUNK=main" "MAPX_PLAN_AFTER_TRUNK=tower" "MAPX_HEAD_DW_LATE=0" "MAPX_TAIL_OVERLAP=0" "MAPX_EARLY_TABLE_UPDATE=1" "MAPX_DW_BATCH=0" "MAPX_CROSS_FUSE=1" "MAPX_X0_LINK=0" "MAPX_X=0"; do
    a=$(env $E python tools/step_bench.py --pt $PT 2>/dev/null | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/')
    b=$(env $E python tools/step_bench.py --pt $PT 2>/dev/null | tail -1 | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/')
    echo "$PT [$E] $a $b"
  done
done
