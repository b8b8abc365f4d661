#!/bin/bash
# On the GPU box: same-box A/B of two libraries over the general-kernel shapes.  ab_shapes.sh LIB_A LIB_B
for a in "--tile 8 --block 512" "--tile 4 --block 512" "--tile 2 --block 128" "--workload cfg5_dense_720" "--workload cfg2_4096_static"; do
  echo "== $a"
  AB_ARGS="--dataset-rows off --repeats 3 $a" bash scratch/ab.sh "$@" | sort | uniq -c | sort -k2 | awk '{print "   ", $0}' || exit 1
done
