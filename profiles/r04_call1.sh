#!/bin/bash
# round 4, GPU call 1: validate + A/B the three round-3 experiment branches, then the queued pipeline-knob sweep.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out/c1
B=profiles/build
( time timeout 200 env REVO_HIP_SO=$B/librevo_hip_var_edtlean.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_zz_deferred_edt.py tests/test_gpu_vo.py -m gpu -x -q ) > gpurun_out/c1/pytest_edtlean.log 2>&1; tail -3 gpurun_out/c1/pytest_edtlean.log
( time timeout 200 env REVO_HIP_SO=$B/librevo_hip_var_nms12.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > gpurun_out/c1/pytest_nms12.log 2>&1; tail -3 gpurun_out/c1/pytest_nms12.log
( time timeout 200 env REVO_HIP_SO=$B/librevo_hip_var_reuse.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker2.py tests/test_gpu_vo.py -m gpu -x -q ) > gpurun_out/c1/pytest_reuse.log 2>&1; tail -3 gpurun_out/c1/pytest_reuse.log
timeout 400 python profiles/ab_bench.py --runs 2 base= edtlean=$B/librevo_hip_var_edtlean.so nms12=$B/librevo_hip_var_nms12.so reuse=$B/librevo_hip_var_reuse.so 2>&1 | tee gpurun_out/c1/ab_branches.txt
timeout 500 python profiles/ab_bench.py --runs 2 \
  base= \
  'c2=REVO_TRACK_CLUSTER=2' \
  'd3=REVO_TRACK_DEPTH=3@--buffers 4 --track-streams 3' \
  'd3c2=REVO_TRACK_DEPTH=3,REVO_TRACK_CLUSTER=2@--buffers 4 --track-streams 3' \
  'd4c2=REVO_TRACK_DEPTH=4,REVO_TRACK_CLUSTER=2@--buffers 5 --track-streams 4' \
  'd3c3=REVO_TRACK_DEPTH=3,REVO_TRACK_CLUSTER=3@--buffers 4 --track-streams 3' \
  'b2=@--build-streams 2' \
  2>&1 | tee gpurun_out/c1/sweep_pipeline_knobs.txt
