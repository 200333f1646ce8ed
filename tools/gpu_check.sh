set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/fin
timeout -k 10 900 python -m pytest tests -q -x -m gpu > gpurun_out/fin/gpu_tests.log 2>&1
tail -3 gpurun_out/fin/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/fin/smoke.log 2>&1
tail -2 gpurun_out/fin/smoke.log
python bench.py > gpurun_out/fin/bench.json 2> gpurun_out/fin/bench.err
tail -c 3000 gpurun_out/fin/bench.json
