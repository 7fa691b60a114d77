#!/bin/bash
# round 4, GPU call 17: k_pyrdown with all its loads requested up front (one memory round trip per thread instead of three)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c17; mkdir -p $O
( timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bit_exact_640 or other_level or unusual or u16 or edge_cases" ) > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log
timeout 200 python profiles/ab_bench.py --runs 1 --args "--steps 60" new= old=profiles/build/librevo_hip_var_prepyr.so new2= 2>&1 | tee $O/ab_pyrdown.txt
