#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log
tail -4 gpurun_out/pytest.log
python bench.py --cpu-baseline off --single-stream-frames 0 --input-cache gpurun_out/inp --steps 2 --warmup 1 --no-overlap > /dev/null 2>&1
rm -rf gpurun_out/ks
rocprofv3 --kernel-trace -d $R/gpurun_out/ks -o ks -- python bench.py --cpu-baseline off --single-stream-frames 0 --input-cache gpurun_out/inp --steps 10 --warmup 2 --no-overlap > gpurun_out/ks.log 2>&1
python profiles/summarize_rocpd.py $(find gpurun_out/ks -name '*.db' | head -1) | grep -E "compact|ccl_out|canny" | cut -c1-120
python bench.py --cpu-baseline off --single-stream-frames 0 --input-cache gpurun_out/inp 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"stages_ms": {[^}]*}'
rm -f gpurun_out/inp*.npz; rm -rf gpurun_out/ks
