# The other bench lines quoted in DESIGN §6, on the build that ships: bf16 mode, Criteo-shaped, and the data-parallel
# step launched exactly as the driver launches N > 1 (torch.distributed.run; one rank here, RCCL group of one).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/fin
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/fin/bench_bf16.json 2> gpurun_out/fin/bench_bf16.err
python bench.py --workload criteo --no-cpu-baseline > gpurun_out/fin/bench_criteo.json 2> gpurun_out/fin/bench_criteo.err
python bench.py --workload criteo --dtype bf16 --no-cpu-baseline > gpurun_out/fin/bench_criteo_bf16.json 2> gpurun_out/fin/bench_criteo_bf16.err
MAPX_FORCE_DP=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
  --master-port 29533 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/fin/bench_dp1.json 2> gpurun_out/fin/bench_dp1.err
for f in bench_bf16 bench_criteo bench_criteo_bf16 bench_dp1; do python tools/show_bench.py gpurun_out/fin/$f.json | head -1; done
