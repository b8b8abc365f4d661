#!/bin/bash
# A/B on one box of bench.py argument sets with one library: each line of stdin = extra arguments
R=$GRAFT_REPO_ROOT; cd $R
while IFS= read -r args; do
  for round in 1 2; do
    timeout -k 10 120 python bench.py --no-cpu-baseline --steps 300 $args > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('[$args]', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
  done
done
