#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for args in "--workload cfg2_4096_static" "--workload cfg2_4096_static --tile 2" "--workload cfg2_4096_static --tile 8" "" "--workload cfg5_dense_720 --steps 100" "--workload cfg1_single_episode --steps 1000"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --cnn none $args > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; continue; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('$args', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()}, d['episode_stats']['episodes_done'])"
done
