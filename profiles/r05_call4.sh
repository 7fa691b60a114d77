#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c4; mkdir -p $O
for v in "sync" "nosync" "nosync novo" "nosync nocoll"; do echo "== $v"; timeout 200 python profiles/census_probe2.py $v 2>&1 | grep -v "^\[W\|amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl"; done | tee $O/census_probe2.txt
