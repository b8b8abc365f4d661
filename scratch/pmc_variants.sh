#!/bin/bash
# VALU / SALU wave-instructions per k_sense launch for each library given (timing-only experiment builds)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for lib in "$@"; do
  out=gpurun_out/pmcv; rm -rf $out; mkdir -p $out
  AUVSIM_LIB=$R/$lib timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --spin-seconds 0 --steps 50 --repeats 1 --dataset-rows off $PMC_ARGS > $out/log 2>&1 || { tail -3 $out/log; exit 1; }
  echo "$lib [$PMC_ARGS]: $(python3 scratch/pmc_summary.py $(find $out -name '*counter_collection.csv') | grep k_sense)"
done
