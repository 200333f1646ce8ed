#!/bin/bash
# Where the K-step of the wave-specialised GEMM (csrc/gemm_ws.hip) spends its time: rebuilds gemm_ws on the GPU box
# with in-kernel stamps (-DMAPX_WS_STAMP) and with phases compiled out (-DMAPX_WS_ABLATE=bits: 1 no DMA / loads,
# 2 no MFMAs, 4 no fragment reads) and prints cycles per K-step and the in-kernel clock.
#   gpurun -- 'bash tools/experiments/gemm_ws/ws_ablate.sh'     (WS_ABLATE_BITS="0 1 2" picks)
set -e
cd ${GRAFT_REPO_ROOT:-.}
D=tools/experiments/gemm_ws
for k in ${WS_ABLATE_BITS:-0 1 2 4 6}; do
  rm -f $D/libgemm_ws.so
  make -C $D EXTRA="-DMAPX_WS_STAMP -DMAPX_WS_ABLATE=$k" > /dev/null 2>&1
  echo "ablate bits $k"
  python3 $D/gemm_ws_bench.py stamp 2>&1 | grep -v amdgpu.ids
done
rm -f $D/libgemm_ws.so
make -C $D > /dev/null 2>&1
