#!/bin/bash
# round 4, GPU call 2: the whole -m gpu suite on the merged main, pipeline shapes at gate depth 2 (more batches in rotation,
# the keyframes' EDT on its own stream), and the timeline of the pipelined step as shipped.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c2; mkdir -p $O
( time timeout 400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -n 8 $O/pytest_gpu.log
timeout 600 python profiles/ab_bench.py --runs 2 \
  base= \
  'b4=@--buffers 4' \
  'b4t3=@--buffers 4 --track-streams 3' \
  'e1=@--edt-streams 1' \
  'b4e1=@--buffers 4 --edt-streams 1' \
  'b4e2=@--buffers 4 --edt-streams 2' \
  'b5e1=@--buffers 5 --edt-streams 1' \
  'b4e1c5=REVO_TRACK_CLUSTER=5@--buffers 4 --edt-streams 1' \
  2>&1 | tee $O/ab_pipeline_shapes.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --no-collective --render-procs 1 --input-cache /tmp/revo_ab_inputs"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_base -o base -- $B --steps 24 --warmup 4 > $R/$O/bench_prof_base.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_b4e1 -o b4e1 -- $B --steps 24 --warmup 4 --buffers 4 --edt-streams 1 > $R/$O/bench_prof_b4e1.log 2>&1
cd $R
for v in base b4e1; do
  db=$(find $O/prof_$v -name '*.db' | head -1)
  echo "== $v $db"
  python profiles/stream_timeline.py $db 12 > $O/step_timeline_$v.txt 2>&1
  python profiles/summarize_rocpd.py $db > $O/kernel_stats_overlapped_$v.csv 2>&1
  python profiles/track_overlap.py $db > $O/track_overlap_$v.txt 2>&1
  head -30 $O/step_timeline_$v.txt
  rm -f $db   # (tens of MB; the text files are what is kept)
done
