#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for args in "--workload cfg1_single_episode --steps 1000" "--workload cfg2_4096_static" "--workload cfg3_65536_moving" "--workload cfg5_dense_720 --cnn none" "--workload cfg5_dense_720 --cnn none --lidar-mode 1" "--workload cfg3_65536_moving --envs 262144 --cnn none"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline $args > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('$args', 'ms/step', round(d['ms_per_step'],4), 'env-steps/s %.3g rays/s %.3g' % (d['value'], d['rays_per_sec']), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done
