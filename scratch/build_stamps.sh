#!/bin/bash
# Diagnostic build of libauvsim whose k_sense records the shader clock at each phase boundary (scratch/stamps2.py reads it).
set -e
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-function -DAUV_PHASE_STAMPS=1 \
  -Iinclude -o scratch/libauvsim_stamps.so gym-auv_amd/csrc/auvsim.hip
# the same with only the two device-wide stamps per wave (scratch/timeline.py): registers and timing of the product kernel
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-function -DAUV_PHASE_STAMPS=2 \
  -Iinclude -o scratch/libauvsim_timeline.so gym-auv_amd/csrc/auvsim.hip
ls -la scratch/libauvsim_stamps.so scratch/libauvsim_timeline.so
