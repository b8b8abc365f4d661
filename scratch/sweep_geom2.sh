#!/bin/bash
# tile / workgroup-size sweep of k_sense for the static (cfg2) and dense (cfg5) workloads
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout -k 10 120 python bench.py --no-cpu-baseline --repeats 3 $1 > gpurun_out/ab.log 2>&1 || { echo "[$1] failed: $(tail -1 gpurun_out/ab.log | cut -c1-150)"; return; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('[$1]', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
for t in 1 2 4 8 16; do for b in 64 128 256; do run "--workload cfg2_4096_static --tile $t --block $b"; done; done
for t in 1 2; do for b in 256 512; do run "--workload cfg5_dense_720 --cnn none --steps 100 --tile $t --block $b"; done; done
