#!/bin/bash
# generic k_sense at the quoted workload: tile x block, to see whether a bigger fixed shape could pay
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout -k 10 120 python bench.py --no-cpu-baseline --repeats 3 --dataset-rows off --spin-seconds 0.3 $1 > gpurun_out/ab.log 2>&1 || { echo "[$1] failed: $(tail -1 gpurun_out/ab.log | cut -c1-150)"; return; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('[$1]', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
run ""
for tb in "4 128" "4 512" "2 128" "2 256" "6 384" "8 256" "8 512" "12 768" "16 512" "16 1024"; do set -- $tb; run "--tile $1 --block $2"; done
