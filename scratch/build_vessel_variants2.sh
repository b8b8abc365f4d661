#!/bin/bash
# timing-only builds of k_vessel on top of the one without integration (norkf): where its remaining ~10 us go
set -e
cd /root/repo
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-function"
base=/tmp/var_norkf
mk() { d=/tmp/var_$1; rm -rf $d; cp -r $base $d; }
S=gym-auv_amd/csrc/auv_sense.hpp
mk v_noer;  sed -i -E "s@^    er\[ER_(COS|SIN|VOX|VOY)\] = .*;@    // knocked out@" /tmp/var_v_noer/$S
mk v_nonav; sed -i 's|^    nav\[\([0-5]\)\] = .*;|    // knocked out|' /tmp/var_v_nonav/$S
mk v_norestart; sed -i 's|bool restart = live \&\& a.do_step \&\& a.auto_reset \&\& a.done\[i\] != 0;|bool restart = false;|' /tmp/var_v_norestart/$S
mk v_empty; sed -i 's|    const bool live = i < V;  // no early exit.*|    const bool live = i < V; if (a.V > 0) return;|' /tmp/var_v_empty/$S
for n in v_noer v_nonav v_norestart v_empty; do
  if diff -q /tmp/var_$n/$S $base/$S > /dev/null; then echo "$n: sed changed nothing"; exit 1; fi
  /opt/rocm/bin/hipcc $F -o scratch/libauvsim_$n.so /tmp/var_$n/gym-auv_amd/csrc/auvsim.hip &
done
wait
ls scratch/libauvsim_v_*.so
