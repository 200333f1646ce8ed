#!/bin/bash
# Regenerates profiles/rNN_* on a GPU box:  gpurun -- 'bash tools/profile_round.sh r01'
# (kernel stats of the graphed and of the serial step, kernel stats of the one-rank RCCL
#  rehearsal of the data-parallel step, and the two PMC passes behind roofline.traffic)
set -e
R=${1:-r01}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O
B="python3 bench.py --steps 30 --warmup 5 --preroll 100 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -o g -- $B > $O/graph.log 2>&1
MAPX_GRAPH=0 MAPX_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -o s -- $B > $O/serial.log 2>&1
MAPX_FORCE_DP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dp -o d -- $B > $O/dp.log 2>&1
P="python3 bench.py --steps 10 --warmup 2 --preroll 60 --no-cpu-baseline"
MAPX_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $P > $O/fetch.log 2>&1
MAPX_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $P > $O/write.log 2>&1
mkdir -p $O/out
cp $(find $O/graph -name '*kernel_stats.csv' | head -1) $O/out/${R}_bench_kernel_stats.csv
cp $(find $O/serial -name '*kernel_stats.csv' | head -1) $O/out/${R}_bench_kernel_stats_serial.csv
cp $(find $O/dp -name '*kernel_stats.csv' | head -1) $O/out/${R}_dp_rehearsal_kernel_stats.csv
python3 tools/pmc_summary.py $(find $O/fetch -name '*counter_collection.csv' | head -1) \
        $(find $O/write -name '*counter_collection.csv' | head -1) $O/out/${R}_pmc_hbm_traffic.json > $O/out/pmc.txt
# keep the merge-back small: only the summaries travel
find $O -mindepth 1 -maxdepth 1 ! -name out ! -name '*.log' -exec rm -rf {} +
ls -la $O/out
