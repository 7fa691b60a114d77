#!/bin/bash
# round 6, GPU call 14: fill-in on the auxiliary stream (REVO_DEFER_FILL=1) and five / six batches in rotation, interleaved A/B; parity of the deferred fill
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c14; mkdir -p $O
( REVO_DEFER_FILL=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_tracker2.py tests/test_gpu_zz_deferred_edt.py -m gpu -x -q ) > $O/pytest_defer_fill.log 2>&1; grep -E "passed|failed" $O/pytest_defer_fill.log
timeout 2000 python profiles/ab_bench.py --runs 2 base= fill=REVO_DEFER_FILL=1 'b5=@--buffers 5' 'fillb5=REVO_DEFER_FILL=1@--buffers 5' 'fillb6=REVO_DEFER_FILL=1@--buffers 6' 2>&1 | grep -v amdgpu.ids | tee $O/ab_defer_fill.txt | tail -8
