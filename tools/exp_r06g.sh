#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== fused"; python tools/diag_steady_ba.py 2>&1 | grep -v amdgpu
echo "== unfused"; DEVO_BA_FUSE_RETRACT=0 python tools/diag_steady_ba.py 2>&1 | grep -v amdgpu
echo "== unfused 1 iter"; ITERS=1 DEVO_BA_FUSE_RETRACT=0 python tools/diag_steady_ba.py 2>&1 | grep -v amdgpu
echo "== generic"; DEVO_BA_GENERIC=1 DEVO_BA_FUSE_RETRACT=0 python tools/diag_steady_ba.py 2>&1 | grep -v amdgpu
