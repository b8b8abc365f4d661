#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace stats of the training kernels (scratch/train_rate.py) and of the
# dataset loggers (scratch/dataset_rate.py); summaries into gpurun_out/prof_aux.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
out=gpurun_out/prof_aux; mkdir -p $out
for job in train_rate dataset_rate; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$job -- python3 scratch/$job.py > $out/$job.log 2>&1 || { tail -5 $out/$job.log; exit 1; }
  cp $(find $out/$job -name "*kernel_stats.csv") $out/${job}_kernel_stats.csv
  rm -rf $out/$job
done
grep -E "k_train|k_fmt|k_sense|k_vessel" $out/*_kernel_stats.csv | cut -c1-220
tail -n 4 $out/train_rate.log; tail -n 4 $out/dataset_rate.log
