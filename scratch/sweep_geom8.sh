#!/bin/bash
# generic kernel at the benchmark shape: tile x block
R=$GRAFT_REPO_ROOT; cd $R
run() { AUVSIM_LIB=$R/scratch/libauvsim_generic.so timeout -k 10 150 python bench.py --no-cpu-baseline --repeats 3 --dataset-rows off --spin-seconds 0.3 $1 > gpurun_out/ab.log 2>&1 || { echo "[$1] failed: $(tail -1 gpurun_out/ab.log | cut -c1-150)"; return; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('[$1]', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
for tb in "4 256" "8 512" "8 384" "8 448" "6 384" "2 128" "4 320" "4 192"; do set -- $tb; run "--tile $1 --block $2"; done
