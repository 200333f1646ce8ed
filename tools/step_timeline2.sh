#!/bin/bash
# Kernel timeline of one replayed RFD / finetune step:  gpurun -- 'bash tools/step_timeline2.sh <tag> <marker kernel> [step_bench args]'
# -> gpurun_out/tl_<tag>/timeline.txt   (marker: mask_rfd for RFD steps, take_rows for finetune steps)
set -e
TAG=$1; MARK=$2; shift; shift
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/tl_$TAG
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace -d $O -o t -- python3 tools/step_bench.py --steps 30 --warmup 5 --preroll 60 "$@" > $O/run.log 2>&1
python3 tools/step_timeline.py $(find $O -name '*_results.db' | head -1) 60 $MARK > $O/timeline.txt
find $O -name '*.db' -delete
tail -1 $O/timeline.txt
