#!/bin/bash
# Quick per-kernel profile of the bench step:  gpurun -- 'bash tools/prof_quick.sh <tag> [bench args]'
# -> gpurun_out/prof_<tag>/{graph,serial}_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
set -e
TAG=$1; shift
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/prof_$TAG
rm -rf $O && mkdir -p $O
B="python3 bench.py --steps 30 --warmup 5 --preroll 100 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -o g -- $B > $O/graph.log 2>&1
MAPX_GRAPH=0 MAPX_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o s -- $B > $O/serial.log 2>&1
cp $(find $O/graph -name '*kernel_stats.csv' | head -1) $O/graph_kernel_stats.csv
cp $(find $O/serial -name '*kernel_stats.csv' | head -1) $O/serial_kernel_stats.csv
rm -rf $O/graph $O/serial
