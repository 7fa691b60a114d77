#!/bin/bash
# round 4, GPU call 3: k_hyst regrouped (2 workgroups per frame instead of one per level and frame), a 512-thread / 144 KB
# k_hyst that fits next to a tracker workgroup, a high-priority build stream, cluster 3, deeper speculation; single-stream knobs.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c3; mkdir -p $O
B=profiles/build
( time timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -x -q ) > $O/pytest_main.log 2>&1; tail -n 4 $O/pytest_main.log
( time timeout 300 env REVO_HIP_SO=$B/librevo_hip_var_h512.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -x -q -k "not distribution" ) > $O/pytest_h512.log 2>&1; tail -n 4 $O/pytest_h512.log
timeout 700 python profiles/ab_bench.py --runs 2 \
  base= \
  r04a=$B/librevo_hip_var_r04a.so \
  h512=$B/librevo_hip_var_h512.so \
  'prio=@--build-priority -1' \
  "h512prio=$B/librevo_hip_var_h512.so@--build-priority -1" \
  'c3=REVO_TRACK_CLUSTER=3' \
  "h512c3=$B/librevo_hip_var_h512.so,REVO_TRACK_CLUSTER=3" \
  'k3344=REVO_TRACK_KSPEC=3344' \
  2>&1 | tee $O/ab_hyst_prio.txt
timeout 200 python profiles/single_stream_sweep.py 2244:16:1024 2244:16:2048 2244:8:2048 2244:32:1024 4:16:1024 3344:16:2048 2>&1 | grep -v amdgpu.ids | tee $O/single_stream_sweep.txt
