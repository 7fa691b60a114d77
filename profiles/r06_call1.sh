#!/bin/bash
# round 6, GPU call 1: the rewritten k_track decision (scalar per-wave LM state, error sums in double, leaner LDLT tail / exp):
# tracker parity tests, the 128-pair distribution against both oracles, phase profile of the batch and of the single pair
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c1; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_tracker2.py tests/test_gpu_variants.py::test_tracker_tolerance_distribution_128_pairs tests/test_gpu_parity.py -m gpu -x -q -s ) > $O/pytest_tracker.log 2>&1
grep -E "passed|failed|error|Error|soak slice|bench config|evals gpu" $O/pytest_tracker.log | head -20
REVO_HIP_SO=profiles/build/librevo_hip_prof.so timeout 300 python profiles/batch_phases.py > $O/batch_phases.txt 2>&1; tail -12 $O/batch_phases.txt
REVO_HIP_SO=profiles/build/librevo_hip_prof.so PH_KSPEC=2244 PH_CLUSTER=16 timeout 300 python profiles/single_pair_phases.py > $O/single_pair_phases.txt 2>&1; tail -4 $O/single_pair_phases.txt
