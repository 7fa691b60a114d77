#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -s > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
timeout 300 python bench.py --input-cache /tmp/revo_in > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-baseline off --no-overlap --single-stream-frames 0 --input-cache /tmp/revo_in"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o pmc -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o pmc -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.log 2>&1; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
grep -E "1280|passed|failed|Error|error" gpurun_out/pytest.log | head -20; ls gpurun_out/pmc_fetch gpurun_out/pmc_write
