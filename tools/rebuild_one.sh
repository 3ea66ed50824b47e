#!/bin/bash
# Debug helper: recompile ONE source of libdevo_hip.so and relink (python -m devo_amd.build recompiles all nine: minutes).
#   tools/rebuild_one.sh gemm_rs [extra hipcc flags]
set -e
src=$1; shift
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F "$@" -c devo_amd/csrc/$src.hip -o devo_amd/lib/$src.o
objs=""
for s in lie corr ba update linear linear_dw mlp2 gemm_rs events; do objs="$objs devo_amd/lib/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o devo_amd/lib/libdevo_hip.so $objs
python - <<'PY'
from devo_amd import build
print(build.build_binding(verbose=False))
PY
