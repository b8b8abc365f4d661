#!/bin/bash
# On the GPU box: kernel-trace stats and PMC passes of one bench workload, summaries into gpurun_out/prof_$1.
#   profile_round.sh TAG [bench.py arguments ...]
tag=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
out=gpurun_out/prof_$tag; mkdir -p $out
B="--no-cpu-baseline --spin-seconds 0 --steps 200 --repeats 1 --dataset-rows off $*"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 bench.py $B > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
cp $(find $out/kt -name "*kernel_stats.csv") $out/kernel_stats.csv
run_pmc() { name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py $B > $out/$name.log 2>&1 || { tail -5 $out/$name.log; exit 1; }
  python3 scratch/pmc_summary.py $(find $out/$name -name "*counter_collection.csv") >> $out/pmc.txt; }
: > $out/pmc.txt
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY || exit 1
run_pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE || exit 1
run_pmc fetch FETCH_SIZE || exit 1
run_pmc write WRITE_SIZE || exit 1
cat $out/pmc.txt; grep -E "k_sense|k_vessel|k_cnn|k_stats" $out/kernel_stats.csv | cut -c1-200
rm -rf $out/kt $out/sq1 $out/sq2 $out/fetch $out/write
