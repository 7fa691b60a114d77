#!/bin/bash
# round 6, GPU call 16: the sequential stream's device timeline on the bench's own sweep (60 frames, 4 levels): kernel trace of the IO-thread driver
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c16; mkdir -p $O
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof_seq -o s -- python profiles/single_stream_profile.py 60 4 > $O/single_stream_profile.raw 2>&1
grep -E "^(io_thread|lookahead|push):" $O/single_stream_profile.raw > $O/single_stream_profile.txt; cat $O/single_stream_profile.txt
python profiles/seq_timeline.py $(find $O/prof_seq -name '*.db' | head -1) 10 3 45 > $O/single_stream_timeline.txt 2>&1; head -40 $O/single_stream_timeline.txt
find $O -name '*.db' -delete
