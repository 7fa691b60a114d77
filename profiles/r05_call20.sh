#!/bin/bash
# round 5, GPU call 20: the depth half of cv::pyrDown + FilterSubsampleWithHoles leaves the build stream with the deferred edge lists
# (k_pyrdown<GRAY, DEPTH>: same threads, same arithmetic, two launches): the whole -m gpu suite, then A/B against the fused launch
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c20; mkdir -p $O
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error|Error" $O/pytest_gpu.log | head -5
timeout 900 python profiles/ab_bench.py --runs 2 split= 'fused=REVO_SPLIT_DEPTH=0' 2>&1 | tee $O/ab_split_depth.txt
timeout 300 python profiles/ab_bench.py --runs 1 --args "--width 1280 --height 960 --levels 5 --steps 30 --warmup 5 --input-cache /tmp/revo_ab_1280" split= 'fused=REVO_SPLIT_DEPTH=0' 2>&1 | tee $O/ab_split_depth_1280.txt
