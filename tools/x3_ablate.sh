#!/bin/bash
# What the K-step of the split-bf16 GEMM spends its time on: rebuilds gemm_x3 on the GPU box with phases of
# the K loop compiled out (-DMAPX_X3_ABLATE=bits: 1 no global loads, 2 no cut / LDS stores / loads,
# 4 no MFMAs; woven loop: 8 no cut of B, 16 no LDS stores, 32 no global loads; X3_ABLATE_BITS="0 16 32" picks) and times 4096 x 1000 x 4096 on both 128 x 128 layouts.   gpurun -- 'bash tools/x3_ablate.sh'
set -e
cd ${GRAFT_REPO_ROOT:-.}
for k in ${X3_ABLATE_BITS:-0 1 2 4 6}; do
  touch map-code_amd/csrc/gemm_x3.hip
  make -C map-code_amd/csrc EXTRA=-DMAPX_X3_ABLATE=$k > /dev/null 2>&1
  echo "ablate bits $k"
  MAPX_GEMM=x3 python3 tools/gemm_f32_bench.py ablate 2>&1 | grep tile
done
