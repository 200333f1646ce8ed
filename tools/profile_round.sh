#!/bin/bash
# Regenerates profiles/rNN_* on a GPU box:  gpurun -- 'bash tools/profile_round.sh r03'
# For each compute dtype (f32, then bf16 with the suffix _bf16): kernel stats of the graphed and of the
# serial step and the two PMC passes behind roofline.traffic; for f32 also the kernel stats of the
# one-rank RCCL rehearsal of the data-parallel step and of the RFD / finetune steps (tools/step_bench.py).
# Copy gpurun_out/prof_rNN/out/* into profiles/.
set -e
R=${1:-r04}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/prof_$R
rm -rf $O && mkdir -p $O/out
for DT in f32 bf16; do
  S=""; [ $DT = bf16 ] && S="_bf16"
  B="python3 bench.py --steps 30 --warmup 5 --preroll 100 --no-cpu-baseline --dtype $DT"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph$S -o g -- $B > $O/graph$S.log 2>&1
  MAPX_GRAPH=0 MAPX_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial$S -o s -- $B > $O/serial$S.log 2>&1
  P="python3 bench.py --steps 10 --warmup 2 --preroll 60 --no-cpu-baseline --dtype $DT"
  MAPX_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch$S -o f -- $P > $O/fetch$S.log 2>&1
  MAPX_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write$S -o w -- $P > $O/write$S.log 2>&1
  MAPX_GRAPH=0 MAPX_SERIAL=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma$S -o m -- $P > $O/mfma$S.log 2>&1
  python3 tools/pmc_mfma_summary.py $(find $O/mfma$S -name '*counter_collection.csv' | head -1) $O/out/${R}_pmc_mfma_busy$S.json > $O/out/mfma$S.txt
  cp $(find $O/graph$S -name '*kernel_stats.csv' | head -1) $O/out/${R}_bench_kernel_stats$S.csv
  cp $(find $O/serial$S -name '*kernel_stats.csv' | head -1) $O/out/${R}_bench_kernel_stats_serial$S.csv
  python3 tools/pmc_summary.py $(find $O/fetch$S -name '*counter_collection.csv' | head -1) \
          $(find $O/write$S -name '*counter_collection.csv' | head -1) $O/out/${R}_pmc_hbm_traffic$S.json > $O/out/pmc$S.txt
  echo "$DT done"
done
B="python3 bench.py --steps 30 --warmup 5 --preroll 100 --no-cpu-baseline"
MAPX_FORCE_DP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dp -o d -- $B > $O/dp.log 2>&1
cp $(find $O/dp -name '*kernel_stats.csv' | head -1) $O/out/${R}_dp_rehearsal_kernel_stats.csv
# the other BASELINE configurations' captured steps (configs[3] RFD, configs[4] finetune): kernel stats inside the graph
for PT in RFD CTR; do
  L=$(echo $PT | tr A-Z a-z)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$L -o p -- python3 tools/step_bench.py --pt $PT --steps 30 --warmup 5 --preroll 100 > $O/$L.log 2>&1
  cp $(find $O/$L -name '*kernel_stats.csv' | head -1) $O/out/${R}_${L}_step_kernel_stats.csv
done
# keep the merge-back small: only the summaries travel
find $O -mindepth 1 -maxdepth 1 ! -name out ! -name '*.log' -exec rm -rf {} +
ls -la $O/out
