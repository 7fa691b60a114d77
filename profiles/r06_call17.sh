#!/bin/bash
# round 6, GPU call 17: quality vote + past-cloud copies on a stream of their own (REVO_VOTE_STREAM, default 1) -- VO tests, then the
# sequential stream A/B (0 = tracker stream as before) through bench.py's own sweep, then its device timeline
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c17; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_vo.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_vo.txt
for rep in 1 2; do for v in 0 1; do
  REVO_VOTE_STREAM=$v timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline off --skip-host-buffers --single-stream-runs 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['single_stream']
print('REVO_VOTE_STREAM=$v rep $rep: value %.0f  single stream %.0f frames/s (%s) runs %s' % (d['value'], s['frames_per_s'], s.get('statistic'), [round(x) for x in s.get('frames_per_s_runs', [])]))"
done; done 2>&1 | tee $O/ab_vote_stream.txt
timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof_seq -o s -- python profiles/single_stream_profile.py 60 4 > $O/single_stream_profile.raw 2>&1
grep -E "^(io_thread|lookahead|push):" $O/single_stream_profile.raw > $O/single_stream_profile.txt; cat $O/single_stream_profile.txt
python profiles/seq_timeline.py $(find $O/prof_seq -name '*.db' | head -1) 10 3 45 > $O/single_stream_timeline.txt 2>&1; head -45 $O/single_stream_timeline.txt
find $O -name '*.db' -delete
