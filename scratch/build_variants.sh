#!/bin/bash
# build ablation variants of corr.hip into scratch/lib_abl<n>.so
cd /root/repo
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -fno-slp-vectorize"
for n in "$@"; do
  ( hipcc $FL -DDEVO_ABL=$n -c devo_amd/csrc/corr.hip -o scratch/corr_abl$n.o && hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/lib_abl$n.so devo_amd/lib/lie.o scratch/corr_abl$n.o devo_amd/lib/ba.o ) &
done
wait
ls -la scratch/*.so
