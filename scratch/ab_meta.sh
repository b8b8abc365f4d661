#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "metadata_refresh or cfg3_moving" > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
for args in "--cnn none" "--cnn none --meta-every 5" "--cnn none --meta-every 0" "--meta-every 5"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline $args --steps 300 > gpurun_out/ab.log 2>&1 || { tail -5 gpurun_out/ab.log; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/ab.log').read().strip().split('\n')[-1]); print('$args', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done
