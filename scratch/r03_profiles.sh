#!/bin/bash
# On the GPU box: the round's committed measurements: rocprofv3 kernel stats + four PMC passes for cfg3, cfg2 and cfg5
# (scratch/profile_round.sh), the phase stamps of cfg5 and cfg3, and the bench lines themselves.
set -e
mkdir -p gpurun_out/r03
bash scratch/profile_round.sh r03_cfg3 > gpurun_out/r03/prof_cfg3.log 2>&1
bash scratch/profile_round.sh r03_cfg2 --workload cfg2_4096_static > gpurun_out/r03/prof_cfg2.log 2>&1
bash scratch/profile_round.sh r03_cfg5 --workload cfg5_dense_720 --steps 100 > gpurun_out/r03/prof_cfg5.log 2>&1
AUVSIM_LIB=scratch/libauvsim_stamps.so timeout -k 10 120 python scratch/stamps2.py cfg5_dense_720 > gpurun_out/r03/phase_stamps_cfg5.txt 2>&1
AUVSIM_LIB=scratch/libauvsim_stamps.so timeout -k 10 120 python scratch/stamps2.py cfg3_65536_moving > gpurun_out/r03/phase_stamps_cfg3.txt 2>&1
timeout -k 10 200 python bench.py > gpurun_out/r03/bench_cfg3.json.log 2> gpurun_out/r03/bench_cfg3.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_cfg3_20steps.json.log 2> gpurun_out/r03/bench_cfg3_20.err
timeout -k 10 200 python bench.py --workload cfg2_4096_static > gpurun_out/r03/bench_cfg2.json.log 2> gpurun_out/r03/bench_cfg2.err
timeout -k 10 300 python bench.py --workload cfg5_dense_720 --steps 100 > gpurun_out/r03/bench_cfg5.json.log 2> gpurun_out/r03/bench_cfg5.err
# the kernel statistics of the bench's own command (its untimed stepping included: several thousand launches at speed)
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/kt_default -- python3 bench.py --no-cpu-baseline > gpurun_out/r03/kt_default.log 2>&1
cp $(find gpurun_out/r03/kt_default -name "*kernel_stats.csv") gpurun_out/r03/kernel_stats_default_cmd.csv
rm -rf gpurun_out/r03/kt_default
tail -c 400 gpurun_out/r03/bench_cfg5.json.log
