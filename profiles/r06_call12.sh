#!/bin/bash
# round 6, GPU call 12: a kernel trace of the pipelined step kept as a database (analysed off the box: what does each tracker grid wait for?)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c12; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --no-collective --render-procs 1 --input-cache /tmp/revo_r6_inputs"
timeout 300 $B --steps 6 --warmup 2 > /dev/null 2>&1
for st in 0 1; do REVO_STAGE_EDGE_DEPTHS=$st timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof_st$st -o p -- $B --steps 40 --warmup 6 > $R/$O/bench_profiled_st$st.log 2>&1; done

cd $R; ls -la $(find $O -name '*.db'); for st in 0 1; do python profiles/stream_busy.py $(find $O/prof_st$st -name "*.db" | head -1) > $O/stream_busy_st$st.txt; head -24 $O/stream_busy_st$st.txt; done; for st in 1 0 1 0; do REVO_STAGE_EDGE_DEPTHS=$st timeout 600 python bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 80 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"stage $st value\", round(d[\"value\"]), round(d[\"ms_per_step\"],4))"; done
