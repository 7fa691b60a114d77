#!/bin/bash
# round 6, GPU call 18: armed look-ahead (the tracker of the frame after next enqueued with its initialisation open, fed from the
# host the moment it is known; REVO_ARM_AHEAD, default 1) -- its test first (bounded waits: a hang here would be a strike), then the
# VO tests, the sequential-stream A/B and the device timeline
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c18; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_gpu_vo.py -m gpu -x -q -s -k armed 2>&1 | tail -5 | tee $O/pytest_armed.txt
grep -q "1 passed" $O/pytest_armed.txt || exit 1
timeout 600 python -m pytest tests/test_gpu_vo.py tests/test_gpu_parity.py tests/test_gpu_tracker2.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_vo.txt
for rep in 1 2; do for v in 0 1; do
  REVO_ARM_AHEAD=$v timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline off --skip-host-buffers --single-stream-runs 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['single_stream']
print('REVO_ARM_AHEAD=$v rep $rep: value %.0f  single stream %.0f frames/s (%s) runs %s' % (d['value'], s['frames_per_s'], s.get('statistic'), [round(x) for x in s.get('frames_per_s_runs', [])]))"
done; done 2>&1 | tee $O/ab_arm_ahead.txt
timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof_seq -o s -- python profiles/single_stream_profile.py 60 4 > $O/single_stream_profile.raw 2>&1
grep -E "^(io_thread|lookahead|push):" $O/single_stream_profile.raw > $O/single_stream_profile.txt; cat $O/single_stream_profile.txt
python profiles/seq_timeline.py $(find $O/prof_seq -name '*.db' | head -1) 10 3 45 > $O/single_stream_timeline.txt 2>&1; head -18 $O/single_stream_timeline.txt
find $O -name '*.db' -delete
