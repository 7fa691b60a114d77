#!/bin/bash
# round 4, GPU call 6: how much of the build may leave the build stream?  REVO_DEFER = 2 (edge lists + EDT) / 3 (hysteresis and
# fill-in too) on one or two aux streams, with the collective on the tracker's stream.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c6; mkdir -p $O
for lv in 2 3; do
( time timeout 300 env REVO_DEFER=$lv python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker2.py tests/test_gpu_zz_deferred_edt.py tests/test_gpu_hostbatch.py tests/test_gpu_variants.py -m gpu -x -q -k "not distribution" ) > $O/pytest_defer$lv.log 2>&1; tail -n 3 $O/pytest_defer$lv.log | head -1
done
timeout 900 python profiles/ab_bench.py --runs 2 \
  base= \
  'd2e1=REVO_DEFER=2@--edt-streams 1 --coll-on-track' \
  'd2e2=REVO_DEFER=2@--edt-streams 2 --coll-on-track' \
  'd3e1=REVO_DEFER=3@--edt-streams 1 --coll-on-track' \
  'd3e2=REVO_DEFER=3@--edt-streams 2 --coll-on-track' \
  'd3e2b4=REVO_DEFER=3@--edt-streams 2 --coll-on-track --buffers 4' \
  'd2e1b4=REVO_DEFER=2@--edt-streams 1 --coll-on-track --buffers 4' \
  'd3e1b4=REVO_DEFER=3@--edt-streams 1 --coll-on-track --buffers 4' \
  'q8d3e2b4=GPU_MAX_HW_QUEUES=8,REVO_DEFER=3@--edt-streams 2 --coll-on-track --buffers 4' \
  2>&1 | tee $O/ab_defer_levels.txt
