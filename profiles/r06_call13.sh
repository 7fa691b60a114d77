#!/bin/bash
# round 6, GPU call 13: instruction-issue priority (s_setprio 1 / 2 / 3) for the build stream's kernels, interleaved A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c13; mkdir -p $O
timeout 1500 python profiles/ab_bench.py --runs 2 base= prio1=profiles/build/librevo_hip_var_prio1.so prio2=profiles/build/librevo_hip_var_prio2.so prio3=profiles/build/librevo_hip_var_prio3.so 2>&1 | grep -v amdgpu.ids | tee $O/ab_build_prio.txt | tail -12
