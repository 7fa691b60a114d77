#!/bin/bash
# round 4, GPU call 4: is the --edt-streams collapse hardware-queue aliasing?  (8 HW queues / the collective on the tracker's
# stream / no collective), the edge lists deferred with the EDT, tracker streams at high priority.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c4; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker2.py tests/test_gpu_zz_deferred_edt.py -m gpu -x -q ) > $O/pytest_main.log 2>&1; tail -n 4 $O/pytest_main.log
( time timeout 300 env REVO_PTS_DEFER=1 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker2.py tests/test_gpu_zz_deferred_edt.py tests/test_gpu_hostbatch.py -m gpu -x -q ) > $O/pytest_ptsdefer.log 2>&1; tail -n 4 $O/pytest_ptsdefer.log
timeout 900 python profiles/ab_bench.py --runs 2 \
  base= \
  'q8=GPU_MAX_HW_QUEUES=8' \
  'q8e1=GPU_MAX_HW_QUEUES=8@--edt-streams 1' \
  'q8e1p=GPU_MAX_HW_QUEUES=8,REVO_PTS_DEFER=1@--edt-streams 1' \
  'q8b4e1p=GPU_MAX_HW_QUEUES=8,REVO_PTS_DEFER=1@--edt-streams 1 --buffers 4' \
  'cte1=@--edt-streams 1 --coll-on-track' \
  'cte1p=REVO_PTS_DEFER=1@--edt-streams 1 --coll-on-track' \
  'nce1p=REVO_PTS_DEFER=1@--edt-streams 1 --no-collective' \
  'p=REVO_PTS_DEFER=1' \
  'tprio=@--track-priority -1' \
  2>&1 | tee $O/ab_queues_defer.txt
