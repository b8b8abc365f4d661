#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
run() { timeout -k 10 150 python bench.py --no-cpu-baseline --repeats 3 --dataset-rows off --spin-seconds 0.3 $1 > gpurun_out/ab.log 2>&1 || { echo "[$1] failed: $(tail -1 gpurun_out/ab.log | cut -c1-150)"; return; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('[$AUVSIM_LIB $1]', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"; }
export AUVSIM_LIB=$R/scratch/libauvsim_generic.so
for tb in "4 256" "8 512" "2 128"; do set -- $tb; run "--tile $1 --block $2"; done
unset AUVSIM_LIB
run ""
run "--workload cfg2_4096_static"
for tb in "2 256" "4 256"; do set -- $tb; run "--workload cfg2_4096_static --tile $1 --block $2"; done
for tb in "1 768" "1 512" "1 1024" "2 1024"; do set -- $tb; run "--workload cfg5_dense_720 --cnn none --steps 100 --tile $1 --block $2"; done
