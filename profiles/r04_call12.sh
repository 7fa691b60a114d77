#!/bin/bash
# round 4, GPU call 12: the hand-run soaks with this round's kernels: 256 tracker pairs vs the oracle; randomised build scenes
# (incl. 1280x1024 and 1920x1080), as shipped and with the banded hysteresis forced
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c12; mkdir -p $O
timeout 400 python tests/tools/soak_gpu_tracker.py 256 1000 2>&1 | grep -v amdgpu.ids | tee $O/parity_soak.txt
timeout 300 python tests/tools/soak_gpu_parity.py 32 2>&1 | grep -v amdgpu.ids | tail -34 | tee $O/build_soak.txt
timeout 300 env REVO_HYST_BANDED=1 python tests/tools/soak_gpu_parity.py 24 77 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/build_soak_banded.txt
