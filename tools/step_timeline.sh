#!/bin/bash
# Kernel timeline of one graph-replayed step:  gpurun -- 'bash tools/step_timeline.sh <tag> [bench args]'
# -> gpurun_out/tl_<tag>/timeline.txt
set -e
TAG=$1; shift
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/tl_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace -d $O -o t -- python3 bench.py --steps 30 --warmup 5 --preroll 100 --no-cpu-baseline "$@" > $O/run.log 2>&1
python3 tools/step_timeline.py $(find $O -name '*_results.db' | head -1) 120 > $O/timeline.txt
find $O -name '*.db' -delete
tail -1 $O/timeline.txt
