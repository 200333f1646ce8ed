#!/bin/bash
# One-box A/B of environment switches on tools/step_bench.py (RFD / CTR steps):
#   gpurun -- 'bash tools/ab_step.sh "--pt RFD" "ENV=a" "ENV=b" ...'
cd ${GRAFT_REPO_ROOT:-.}
ARGS=$1; shift
for rep in 1 2; do
  for E in "$@"; do
    echo "rep $rep [$E] $(env $E python3 tools/step_bench.py $ARGS 2>/dev/null | tail -1 | cut -c1-170)"
  done
done
