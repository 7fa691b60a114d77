#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
timeout 200 python bench.py --input-cache /tmp/revo_in --cpu-baseline off --single-stream-frames 0 --no-overlap --steps 10 > gpurun_out/bench_exp.log 2>&1
grep -E "passed|failed" gpurun_out/pytest.log
