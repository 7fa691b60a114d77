#!/bin/bash
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "evals gpu|passed|failed|Error|error" | head
python bench.py --cpu-baseline off --single-stream-frames 0 --input-cache gpurun_out/inp --steps 2 --warmup 1 > /dev/null 2>&1
rm -rf gpurun_out/ks
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks -o ks -- python bench.py --cpu-baseline off --single-stream-frames 0 --input-cache gpurun_out/inp --steps 10 --warmup 3 > gpurun_out/ks.log 2>&1
DB=$(find gpurun_out/ks -name '*.db' | head -1)
python profiles/overlap_slowdown.py $DB > gpurun_out/r01_overlap_v9.txt
tail -1 gpurun_out/ks.log > gpurun_out/r01_bench_v9_profiled.log
rm -rf gpurun_out/ks
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks -o ks -- python bench.py --cpu-baseline off --single-stream-frames 0 --input-cache gpurun_out/inp --steps 10 --warmup 3 --no-overlap > gpurun_out/ks.log 2>&1
python profiles/summarize_rocpd.py $(find gpurun_out/ks -name '*.db' | head -1) > gpurun_out/r01_kernel_stats_v9.csv
rm -f gpurun_out/inp*.npz; rm -rf gpurun_out/ks
python bench.py 2>/dev/null | tail -1 > gpurun_out/r01_bench_v9.log
cut -c1-300 gpurun_out/r01_bench_v9.log
