#!/bin/bash
# What the K-step of the two-piece fp16 GEMM spends its time on: rebuilds gemm_h2 on the GPU box with phases of the
# woven K loop compiled out (-DMAPX_H2_ABLATE=bits: 2 no cut / LDS stores / global loads, 4 no MFMAs, 16 no LDS
# stores, 32 no global loads; results are then wrong) and times 4096 x 1000 x K.   gpurun -- 'bash tools/h2_ablate.sh'
# H2_ABLATE_W8=1: the same for the 8-wave weight-planes kernel of gemm_h2w.hip (-DMAPX_W8_ABLATE=bits: 1 no B loads,
# 2 no cut, 4 no MFMAs, 8 no LDS stores, 16 no A loads, 32 no fragment reads, 64 no barrier).
set -e
cd ${GRAFT_REPO_ROOT:-.}
if [ -n "$H2_ABLATE_W8" ]; then
  for k in ${H2_ABLATE_BITS:-0 1 2 4 8 16 32 64}; do
    touch map-code_amd/csrc/gemm_h2w.hip
    make -C map-code_amd/csrc EXTRA=-DMAPX_W8_ABLATE=$k > /dev/null 2>&1
    echo "w8 ablate bits $k"
    python3 tools/gemm_h2_bench.py ablatew 2>&1 | grep "K="
  done
  touch map-code_amd/csrc/gemm_h2w.hip
  make -C map-code_amd/csrc > /dev/null 2>&1
  exit 0
fi
for k in ${H2_ABLATE_BITS:-0 2 4 16 32 48}; do
  touch map-code_amd/csrc/gemm_h2.hip
  make -C map-code_amd/csrc EXTRA=-DMAPX_H2_ABLATE=$k > /dev/null 2>&1
  echo "ablate bits $k"
  python3 tools/gemm_h2_bench.py ablate 2>&1 | grep "K="
done
touch map-code_amd/csrc/gemm_h2.hip
make -C map-code_amd/csrc > /dev/null 2>&1
