#!/bin/bash
# round 5, GPU call 22 (the depth half of pyrDown on the auxiliary stream): rocprofv3 evidence on the final sources (commit c0eb6ab): kernel stats alone and in the pipelined step (the
# library-owned pipeline), step timeline, tracker overlap, PMC passes (FETCH_SIZE / WRITE_SIZE separately; wait + instruction counters)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c22; mkdir -p $O
R=$GRAFT_REPO_ROOT
export REVO_COMMIT=c0eb6ab
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --no-collective --render-procs 1 --input-cache /tmp/revo_r5_inputs"
timeout 300 $B --steps 6 --warmup 2 > /dev/null 2>&1   # renders the inputs once (the profiler deadlocks on the render pool)
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_alone -o p -- $B --steps 10 --warmup 3 --no-overlap > $R/$O/bench_profiled_alone.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- $B --steps 24 --warmup 4 > $R/$O/bench_profiled_overlapped.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- $B --steps 6 --warmup 3 > $R/$O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- $B --steps 6 --warmup 3 > $R/$O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $R/$O/pmc_wait -o q -- $B --steps 5 --warmup 2 --no-overlap > $R/$O/pmc_wait.log 2>&1
cd $R
db() { find $O/$1 -name '*.db' | head -1; }
python profiles/summarize_rocpd.py $(db prof_alone) > $O/kernel_stats.csv 2>&1
python profiles/summarize_rocpd.py $(db prof) > $O/kernel_stats_overlapped.csv 2>&1
python profiles/stream_timeline.py $(db prof) 12 > $O/step_timeline.txt 2>&1
python profiles/track_overlap.py $(db prof) > $O/track_overlap.txt 2>&1
python profiles/overlap_slowdown.py $(db prof) > $O/overlap.txt 2>&1
python profiles/pmc_summary.py $(db pmc_fetch) $(db pmc_write) 32 640 480 "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- $B --steps 6 --warmup 3" borrow > $O/pmc_summary.json 2> $O/pmc_summary.err
python profiles/pmc_by_kernel.py $(db pmc_wait) k_ > $O/pmc_wait_and_instructions.txt 2>&1
cat $O/step_timeline.txt | head -30
head -16 $O/kernel_stats.csv | cut -c1-160
tail -3 $O/bench_profiled_overlapped.log | cut -c1-300
find $O -name '*.db' -delete
