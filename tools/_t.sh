cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/x
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_trainer_gpu.py -q -x -m gpu -k "xDeepFM or unknown_backbones" > gpurun_out/x/t.log 2>&1
echo rc=$?
tail -40 gpurun_out/x/t.log
