#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for args in "--tile 4" "--tile 3" "--tile 5" "--tile 6" "--tile 2" "--tile 8 --block 512"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --cnn none $args --steps 200 > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('$args', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done
