#!/bin/bash
# round 5, GPU call 21: step timeline with the depth half of pyrDown on the auxiliary stream
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c21; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --no-collective --render-procs 1 --input-cache /tmp/revo_r5_inputs"
timeout 300 $B --steps 6 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- $B --steps 24 --warmup 4 > $R/$O/bench_profiled_overlapped.log 2>&1
cd $R
db() { find $O/$1 -name '*.db' | head -1; }
python profiles/stream_timeline.py $(db prof) 12 > $O/step_timeline.txt 2>&1
python profiles/stream_timeline.py $(db prof) 17 > $O/step_timeline_b.txt 2>&1
python profiles/overlap_slowdown.py $(db prof) > $O/overlap.txt 2>&1
cat $O/step_timeline.txt; cat $O/step_timeline_b.txt | head -24; cat $O/overlap.txt
find $O -name '*.db' -delete
