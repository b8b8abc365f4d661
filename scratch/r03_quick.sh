#!/bin/bash
# On the GPU box: parity tests of the sensing kernels, then short bench lines of cfg5, cfg2 and cfg3.  r03_quick.sh TAG
tag=$1; out=gpurun_out/r03; mkdir -p $out
timeout -k 10 500 python -u -m pytest tests/test_parity_gpu.py tests/test_bounds_gpu.py tests/test_stats_gpu.py tests/test_ref_contact_gpu.py -x -q --timeout 180 --timeout-method thread > $out/par_$tag.log 2>&1 || { tail -30 $out/par_$tag.log; exit 1; }
tail -2 $out/par_$tag.log
B="--no-cpu-baseline --steps 100 --repeats 3 --dataset-rows off"
timeout -k 10 150 python bench.py $B --workload cfg5_dense_720 > $out/b5_$tag.log 2>&1 || { tail -5 $out/b5_$tag.log; exit 1; }
timeout -k 10 150 python bench.py $B --workload cfg2_4096_static > $out/b2_$tag.log 2>&1 || { tail -5 $out/b2_$tag.log; exit 1; }
timeout -k 10 150 python bench.py $B > $out/b3_$tag.log 2>&1 || { tail -5 $out/b3_$tag.log; exit 1; }
python - <<PY
import json
for w in ("b5","b2","b3"):
    d=json.loads([l for l in open("$out/%s_$tag.log" % w) if l.startswith("{")][-1])
    print(w, "ms/step %.4f" % d["ms_per_step"], d["kernel_ms"], "frac %.3f" % d["roofline"]["frac"])
PY
