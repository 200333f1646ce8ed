cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dp
export MAPX_DIST_BACKEND=gloo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 10 --warmup 2 --preroll 10 --rows 400000 --no-cpu-baseline > gpurun_out/dp/gloo2.json 2> gpurun_out/dp/gloo2.err
echo rc=$?
tail -c 600 gpurun_out/dp/gloo2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/dp/gloo2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['n_gpus'], d['final_loss'], d.get('dp'))
PY
