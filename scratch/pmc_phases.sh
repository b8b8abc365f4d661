#!/bin/bash
# run on the GPU box: per-variant VALU/SALU/LDS instruction counts and kernel time of k_sense
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R; export PYTHONPATH=$R
for v in stop1 stop2 stop4 stop5 stop7 full; do
  if [ $v = full ]; then export AUVSIM_LIB=$R/gym-auv_amd/libauvsim.so; else export AUVSIM_LIB=$R/scratch/libauvsim_$v.so; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pmc_$v --output-format csv -- python3 scratch/run_sense.py 6 > gpurun_out/pmc_$v.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d gpurun_out/kt_$v --output-format csv -- python3 scratch/run_sense.py 30 > gpurun_out/kt_$v.log 2>&1 || exit 1
  echo "== $v"
  python3 scratch/pmc_summary.py $(find gpurun_out/pmc_$v -name "*counter_collection.csv") | grep k_sense
  grep k_sense $(find gpurun_out/kt_$v -name "*kernel_stats.csv") | cut -d, -f1-6 | cut -c1-20,60-
done
