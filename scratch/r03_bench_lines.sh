#!/bin/bash
# On the GPU box: the committed bench lines (after scratch/collect_profiles.py has refreshed profiles/traffic.json).
set -e
mkdir -p gpurun_out/r03
timeout -k 10 200 python bench.py > gpurun_out/r03/bench_cfg3.json.log 2> gpurun_out/r03/bench_cfg3.err
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_cfg3_20steps.json.log 2> gpurun_out/r03/bench_cfg3_20.err
timeout -k 10 200 python bench.py --workload cfg2_4096_static > gpurun_out/r03/bench_cfg2.json.log 2> gpurun_out/r03/bench_cfg2.err
timeout -k 10 300 python bench.py --workload cfg5_dense_720 --steps 100 > gpurun_out/r03/bench_cfg5.json.log 2> gpurun_out/r03/bench_cfg5.err
AUV_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline --envs 16384 > gpurun_out/r03/bench_2rank_rehearsal.json.log 2> gpurun_out/r03/bench_2rank.err
AUV_BENCH_FORCE_GROUP=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03/bench_cfg3_rccl_group.json.log 2> gpurun_out/r03/bench_rccl.err
python - <<'PY'
import json
for c in ("cfg3","cfg3_20steps","cfg2","cfg5","2rank_rehearsal","cfg3_rccl_group"):
    d=json.loads([x for x in open(f"gpurun_out/r03/bench_{c}.json.log") if x.startswith("{")][-1])
    print(c, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], d["kernel_ms"], "frac %.3f"%d["roofline"]["frac"], d.get("collectives",{}).get("backend"))
PY
