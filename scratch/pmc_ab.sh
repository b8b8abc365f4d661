#!/bin/bash
# On the GPU box: instruction counters of k_sense for several libraries on one bench shape.  pmc_ab.sh "BENCH ARGS" LIB...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
args=$1; shift
for lib in "$@"; do
  d=gpurun_out/pmc_ab; rm -rf $d; mkdir -p $d
  AUVSIM_LIB=$R/$lib timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $d -- python3 bench.py --no-cpu-baseline --spin-seconds 0 --steps 100 --repeats 1 --dataset-rows off $args > $d/log.txt 2>&1 || { tail -5 $d/log.txt; exit 1; }
  echo "$lib [$args]: $(python3 scratch/pmc_summary.py $(find $d -name '*counter_collection.csv') | grep k_sense)"
done
