#!/bin/bash
# gemm_h2's K-step with N units of the cut ahead of the MFMAs instead of woven between them (-DMAPX_H2_PRE=N)
set -e
cd ${GRAFT_REPO_ROOT:-.}
for k in ${H2_PRE:-4 12 20 31}; do
  touch map-code_amd/csrc/gemm_h2.hip
  make -C map-code_amd/csrc EXTRA=-DMAPX_H2_PRE=$k > /dev/null 2>&1
  echo "pre units $k"
  python3 tools/gemm_h2_bench.py ablate 2>&1 | grep "K="
done
touch map-code_amd/csrc/gemm_h2.hip
make -C map-code_amd/csrc > /dev/null 2>&1
