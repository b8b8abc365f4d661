#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for wl in "cfg2_4096_static" "cfg1_single_episode"; do
for args in "--tile 1" "--tile 2" "--tile 4" "--tile 8" "--tile 1 --block 64" "--tile 2 --block 128" "--tile 1 --block 128" "--tile 4 --block 128"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --cnn none --workload $wl $args --steps 500 > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('$wl $args', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done; done
