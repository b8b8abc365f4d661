#!/bin/bash
# experiment builds of libauvsim from sed-edited copies of the sources: name|sed-expression
set -e
cd /root/repo
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-function"
i=0
while IFS='|' read -r name expr; do
  [ -z "$name" ] && continue
  d=/tmp/var_$name; rm -rf $d; mkdir -p $d/gym-auv_amd $d/include
  cp -r gym-auv_amd/csrc $d/gym-auv_amd/; cp include/auvsim.h $d/include/
  sed -i "$expr" $d/gym-auv_amd/csrc/auv_sense.hpp
  if diff -q $d/gym-auv_amd/csrc/auv_sense.hpp gym-auv_amd/csrc/auv_sense.hpp >/dev/null; then echo "variant $name: sed changed nothing"; exit 1; fi
  /opt/rocm/bin/hipcc $F -o scratch/libauvsim_$name.so $d/gym-auv_amd/csrc/auvsim.hip &
done
wait
ls scratch/*.so
