set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dp
timeout -k 10 500 python -m pytest tests/test_trainer_gpu.py -q -x -m gpu -k "one_rank_rccl" > gpurun_out/dp/test.log 2>&1 || { tail -40 gpurun_out/dp/test.log; exit 1; }
tail -2 gpurun_out/dp/test.log
export MAPX_FORCE_DP=1
python bench.py --steps 200 --warmup 20 --preroll 200 --no-cpu-baseline > gpurun_out/dp/bench_dp.json 2> gpurun_out/dp/bench_dp.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/dp/bench_dp.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('dp'))
PY
