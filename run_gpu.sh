#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
timeout 300 python bench.py --input-cache /tmp/revo_in > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-baseline off --no-overlap --single-stream-frames 0 --input-cache /tmp/revo_in > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1; echo "prof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/pytest.log | cut -c1-300; tail -2 gpurun_out/smoke.log
