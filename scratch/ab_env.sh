#!/bin/bash
# A/B on one box of one library under different values of an environment variable: ab_env.sh VAR v1 v2 ... (3 alternating rounds)
R=$GRAFT_REPO_ROOT; cd $R; var=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    env $var=$v timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 --repeats 3 $AB_ARGS > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('$var=$v', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
  done
done
