#!/bin/bash
# A/B on one box: k_sense time of each library given as argument (paths relative to the repo), 3 alternating rounds
R=$GRAFT_REPO_ROOT; cd $R
for round in 1 2 3; do
  for lib in "$@"; do
    AUVSIM_LIB=$R/$lib timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 $AB_ARGS > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('$lib', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
  done
done
