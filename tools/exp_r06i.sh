#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_binding_legs.py -x -q -m gpu 2>&1 | tail -8
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_binding_legs.py -k "not bench" --co -q 2>/dev/null | tail -1
timeout 600 python tools/profile_train_sections.py 2>&1 | grep -v "amdgpu\|Warning\|warn" | tail -14
