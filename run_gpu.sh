#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-baseline off > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log 2>&1; echo "prof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.log
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/smoke.log; tail -25 gpurun_out/pytest.log | cut -c1-300; tail -2 gpurun_out/bench.log | cut -c1-1500; find gpurun_out/prof -type f | head
