#!/bin/bash
# round 5, GPU call 9: why is the flag phase of k_hyst 200 k cycles on three of the 64 bench frames?  (finer markers: E rebuild /
# flag loop / output / histogram; touched runs and atomics per frame)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c9; mkdir -p $O
REVO_HIP_SO=profiles/build/librevo_hip_var_hp.so timeout 120 python profiles/hyst_profile.py 2>&1 | grep "^hyst f=" | tail -64 | sort -t= -k2 -n > $O/hyst_profile_flag.txt
sort -t' ' -k5 -n -r $O/hyst_profile_flag.txt | head -12
