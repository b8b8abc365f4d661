#!/bin/bash
# On the GPU box: two PMC passes over bench.py with the deep network as a launch of its own; k_cnn_deep's counters.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
out=gpurun_out/prof_cnn_deep; mkdir -p $out; : > $out/pmc.txt
B="--no-cpu-baseline --spin-seconds 0 --steps 100 --repeats 1 --dataset-rows off --cnn deep --cnn-mode separate"
run_pmc() { name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py $B > $out/$name.log 2>&1 || { tail -5 $out/$name.log; exit 1; }
  python3 scratch/pmc_summary.py $(find $out/$name -name "*counter_collection.csv") >> $out/pmc.txt; rm -rf $out/$name; }
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY || exit 1
run_pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA || exit 1
grep -i "cnn" $out/pmc.txt
