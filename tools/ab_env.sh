#!/bin/bash
# A/B of environment switches on ONE box:  gpurun -- 'bash tools/ab_env.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...'
# Runs bench.py (no CPU baseline; extra arguments in $BENCH_ARGS) once per environment string, twice
# round-robin, and prints ms/step (and the serial average us of the kernel classes named in $KERNELS).  The pool's boxes differ by +-4 %: only arms run on one box compare.
TAG=$1; shift
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/ab_$TAG
for rep in 1 2; do
  i=0
  for E in "$@"; do
    i=$((i+1))
    env $E python bench.py --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_$TAG/r${rep}_$i.json 2> gpurun_out/ab_$TAG/r${rep}_$i.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$TAG/r${rep}_$i.json").read().strip().splitlines()[-1])
ks = " ".join("%s %.1f" % (k, d["kernels"][k]["avg_us"]) for k in "${KERNELS}".split() if k in d.get("kernels", {}))
print("rep $rep [$E]  ms/step %.4f  %s" % (d["ms_per_step"], ks), flush=True)
PY
  done
done
