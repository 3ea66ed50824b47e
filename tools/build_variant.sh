#!/bin/bash
# Debug helper: build devo_amd/lib/libdevo_<tag>.so from the current tree with extra compiler flags on ONE source
# (e.g.  tools/build_variant.sh nomath corr -DDEVO_MFMA_DBG_NOMATH), to A/B kernels inside one gpurun call with
# tools/bench_with_lib.py (DEVO_LIB=devo_amd/lib/libdevo_<tag>.so).
set -e
tag=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
python -m devo_amd.build > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F "$@" -c devo_amd/csrc/$src.hip -o devo_amd/lib/${src}_$tag.o
objs=""
for s in lie corr ba update linear linear_dw mlp2 gemm_rs events; do if [ $s = $src ]; then objs="$objs devo_amd/lib/${src}_$tag.o"; else objs="$objs devo_amd/lib/$s.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o devo_amd/lib/libdevo_$tag.so $objs
echo devo_amd/lib/libdevo_$tag.so
