#!/bin/bash
# the three benchmarked workloads at their automatic geometry
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout -k 10 150 python bench.py --no-cpu-baseline --repeats 3 --dataset-rows off --spin-seconds 0.3 $1 > gpurun_out/ab.log 2>&1 || { echo "[$1] failed: $(tail -1 gpurun_out/ab.log | cut -c1-150)"; return; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('[$1]', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
run ""
run "--workload cfg2_4096_static"
run "--workload cfg5_dense_720 --cnn none --steps 100"
