#!/bin/bash
# round 4, GPU call 7: around the new default shape (REVO_DEFER=2, 4 batches in rotation, 1 aux stream, gather on the tracker stream)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c7; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker2.py tests/test_gpu_zz_deferred_edt.py tests/test_gpu_hostbatch.py tests/test_gpu_multi.py tests/test_gpu_vo.py -m gpu -x -q ) > $O/pytest_default.log 2>&1; grep -E "passed|failed|error" $O/pytest_default.log
timeout 900 python profiles/ab_bench.py --runs 2 \
  base= \
  'b3=@--buffers 3' \
  'b5=@--buffers 5' \
  'b6=@--buffers 6' \
  'b5t3=@--buffers 5 --track-streams 3' \
  'd1=REVO_DEFER=1' \
  'nc=@--no-collective' \
  'c3b5=REVO_TRACK_CLUSTER=3@--buffers 5' \
  'c5=REVO_TRACK_CLUSTER=5' \
  2>&1 | tee $O/ab_around_default.txt
