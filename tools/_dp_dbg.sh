cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dp
export PYTHONPATH=$PWD:$PWD/map-code_amd:$PWD/tests:$PWD/tests/golden
MAPX_DP_SPLIT=1 MAPX_FORCE_DP=1 timeout -k 10 300 python tests/dp_rehearsal_worker.py /tmp/x.pt graph > gpurun_out/dp/worker.log 2>&1
rc=$?
echo rc=$rc
grep -v "^\s*$" gpurun_out/dp/worker.log | grep -v "HIP kernel errors\|For debugging\|Compile with\|^frame #" | head -40
