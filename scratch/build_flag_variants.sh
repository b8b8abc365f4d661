#!/bin/bash
# libauvsim built with extra compiler flags, for A/B: name|flags per line on stdin
cd /root/repo
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-function -Iinclude"
while IFS='|' read -r name flags; do
  [ -z "$name" ] && continue
  ( /opt/rocm/bin/hipcc $F $flags -o scratch/libauvsim_$name.so gym-auv_amd/csrc/auvsim.hip 2> /tmp/flag_$name.err || echo "$name: build failed: $(tail -2 /tmp/flag_$name.err)"
    /opt/rocm/bin/hipcc $F $flags -S --cuda-device-only -o /tmp/flag_$name.s gym-auv_amd/csrc/auvsim.hip 2>/dev/null
    python3 - "$name" <<'PY'
import re,sys
n=sys.argv[1]
try:
    s=open(f'/tmp/flag_{n}.s').read()
    m=re.search(r'\.name:\s+(_ZN3auv15k_sense_defaultILi1EEEvNS_9SenseArgsE)\n(.*?)\.wavefront_size', s, re.S)
    b=m.group(2)
    print(n,{k:re.search(r'\.%s:\s+(\d+)'%k, b).group(1) for k in ('sgpr_count','sgpr_spill_count','vgpr_count','vgpr_spill_count')})
except Exception as e: print(n,'no asm',e)
PY
  ) &
done
wait
