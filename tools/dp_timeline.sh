set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/dp
timeout -k 10 500 python -m pytest tests/test_trainer_gpu.py -q -x -m gpu -k "one_rank_rccl or graph_replay" > gpurun_out/dp/test.log 2>&1 || { tail -40 gpurun_out/dp/test.log; exit 1; }
tail -2 gpurun_out/dp/test.log
export MAPX_FORCE_DP=1
python bench.py --steps 100 --warmup 10 --preroll 100 --no-cpu-baseline > gpurun_out/dp/bench_dp.json 2> gpurun_out/dp/bench_dp.err
rocprofv3 --kernel-trace -d gpurun_out/dp -o dp -- python3 bench.py --steps 30 --warmup 5 --preroll 60 --no-cpu-baseline > gpurun_out/dp/prof.log 2>&1
python tools/step_timeline.py gpurun_out/dp/dp_results.db 70 > gpurun_out/dp/timeline.txt
rm -f gpurun_out/dp/dp_results.db
