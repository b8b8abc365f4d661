#!/bin/bash
# timing-only builds of k_vessel: (nosc) the six sincos of the Fehlberg stages replaced by two multiplications,
# (norkf) no integration at all.  They bound what a shorter sincos chain could gain.
set -e
cd /root/repo
F="--offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -shared -Wno-unused-function"
mk() { d=/tmp/var_$1; rm -rf $d; mkdir -p $d/gym-auv_amd $d/include; cp -r gym-auv_amd/csrc $d/gym-auv_amd/; cp include/auvsim.h $d/include/; }
mk nosc
sed -i 's|const SinCos sc = sincos_det(s.psi);\(.*\)|SinCos sc; sc.s = s.psi * 0.5; sc.c = s.psi * 0.25;|' /tmp/var_nosc/gym-auv_amd/csrc/auv_kernels.hpp
sed -n 57p /tmp/var_nosc/gym-auv_amd/csrc/auv_kernels.hpp
mk norkf
sed -i 's|s = rkf45_step(s, a.action\[i\], a.action\[(size_t)V + i\], a.dt, a.thrust_max, a.moment_max, a.mp);|s.x = s.x + a.action[i]; s.psi = s.psi + a.action[(size_t)V + i];|' /tmp/var_norkf/gym-auv_amd/csrc/auv_sense.hpp
grep -n "s.x = s.x + a.action" /tmp/var_norkf/gym-auv_amd/csrc/auv_sense.hpp
for n in nosc norkf; do /opt/rocm/bin/hipcc $F -o scratch/libauvsim_$n.so /tmp/var_$n/gym-auv_amd/csrc/auvsim.hip & done
wait
ls -la scratch/libauvsim_nosc.so scratch/libauvsim_norkf.so
