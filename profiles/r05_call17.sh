#!/bin/bash
# round 5, GPU call 17: k_hyst with 2 / 6 KB less LDS, so that one of its workgroups FITS next to a tracker workgroup on a CU
# (44 VGPRs x 16 waves already fit; 159.3 KB of LDS + the tracker's 2.4 KB did not) -- same kernel otherwise
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c17; mkdir -p $O
REVO_HIP_SO=profiles/build/librevo_hip_var_lds156.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python profiles/ab_bench.py --runs 2 base= lds156=profiles/build/librevo_hip_var_lds156.so lds152=profiles/build/librevo_hip_var_lds152.so 2>&1 | tee $O/ab_hyst_lds.txt
