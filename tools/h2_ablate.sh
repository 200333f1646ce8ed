#!/bin/bash
# What the K-step of the two-piece fp16 GEMM spends its time on: rebuilds gemm_h2 on the GPU box with phases of the
# woven K loop compiled out (-DMAPX_H2_ABLATE=bits: 2 no cut / LDS stores / global loads, 4 no MFMAs, 16 no LDS
# stores, 32 no global loads; results are then wrong) and times 4096 x 1000 x K.   gpurun -- 'bash tools/h2_ablate.sh'
set -e
cd ${GRAFT_REPO_ROOT:-.}
for k in ${H2_ABLATE_BITS:-0 2 4 16 32 48}; do
  touch map-code_amd/csrc/gemm_h2.hip
  make -C map-code_amd/csrc EXTRA=-DMAPX_H2_ABLATE=$k > /dev/null 2>&1
  echo "ablate bits $k"
  python3 tools/gemm_h2_bench.py ablate 2>&1 | grep "K="
done
touch map-code_amd/csrc/gemm_h2.hip
make -C map-code_amd/csrc > /dev/null 2>&1
