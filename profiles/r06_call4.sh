#!/bin/bash
# round 6, GPU call 4: native RCCL collective tests (python + C++ host), pipeline tests, and a speculation-depth / cluster sweep of the single pair
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c4; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_multi.py -m gpu -x -q ) > $O/pytest_pipeline.log 2>&1; tail -15 $O/pytest_pipeline.log
for ks in 2244 3344 4444 2444 2233; do PH_KSPEC=$ks PH_CLUSTER=8,12,16,24 timeout 300 python profiles/single_pair_phases.py 2>&1 | grep "^kspec" ; done | tee $O/single_pair_sweep.txt
