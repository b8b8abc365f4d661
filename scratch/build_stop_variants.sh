#!/bin/bash
# Diagnostic builds of libauvsim whose k_sense returns at a phase boundary.
set -e
cd /root/repo
for n in 1 2 4 5 7; do
  d=/tmp/var_$n; rm -rf $d; mkdir -p $d/gym-auv_amd $d/include
  cp -r gym-auv_amd/csrc $d/gym-auv_amd/; cp include/auvsim.h $d/include/
  sed -i "s/AUV_STAMP($n);/AUV_STAMP($n); return;/" $d/gym-auv_amd/csrc/auv_sense.hpp
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-function \
     -o scratch/libauvsim_stop$n.so $d/gym-auv_amd/csrc/auvsim.hip &
done
wait
ls -la scratch/*.so
