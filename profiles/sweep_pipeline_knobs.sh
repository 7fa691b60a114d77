#!/bin/bash
# One gpurun call: the pipeline-shape experiments that were queued at the end of round 3 and never measured
# (DESIGN section 8).  ~3.5 minutes on the box.  Why depth 3 is the first thing to try: with the keyframes' EDT on the tracker
# streams a batch costs its tracker stream ~97 + 530 us, two streams sustain one batch per ~315 us -- the same as the build
# stream's chain (~313 us) -- so a third tracker stream is what un-binds the step if the CUs have room.  Output: gpurun_out/sweep_pipeline_knobs.txt
#   gpurun --timeout 400 -- 'bash profiles/sweep_pipeline_knobs.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
python profiles/ab_bench.py --runs 2 \
  base= \
  'c2=REVO_TRACK_CLUSTER=2' \
  'd3=REVO_TRACK_DEPTH=3@--buffers 4 --track-streams 3' \
  'd4=REVO_TRACK_DEPTH=4@--buffers 5 --track-streams 4' \
  'd3c3=REVO_TRACK_DEPTH=3,REVO_TRACK_CLUSTER=3@--buffers 4 --track-streams 3' \
  'd3c4=REVO_TRACK_DEPTH=3,REVO_TRACK_CLUSTER=4@--buffers 4 --track-streams 3' \
  'b2=@--build-streams 2' \
  'b2d3=REVO_TRACK_DEPTH=3@--buffers 5 --track-streams 3 --build-streams 2' \
  2>&1 | tee gpurun_out/sweep_pipeline_knobs.txt
