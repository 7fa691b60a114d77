#!/bin/bash
# round 4, GPU call 5: the whole -m gpu suite (new: 1280x1024, 1920x1080, dense 1280x1024), two tracker knobs, and the round's
# profile evidence on the shipped pipeline: kernel stats alone, PMC passes (HBM bytes; wait / instruction counters).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c5; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -n 6 $O/pytest_gpu.log
timeout 400 python profiles/ab_bench.py --runs 2 \
  base= \
  'r1500=REVO_TRACK_REDUNDANT_BATCH=1500' \
  'k2234=REVO_TRACK_KSPEC=2234' \
  'cte1p=REVO_PTS_DEFER=1@--edt-streams 1 --coll-on-track' \
  2>&1 | tee $O/ab_tracker_knobs.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --no-collective --render-procs 1 --input-cache /tmp/revo_ab_inputs"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_alone -o alone -- $B --steps 10 --warmup 3 --no-overlap > $R/$O/bench_prof_alone.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- $B --steps 6 --warmup 3 > $R/$O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- $B --steps 6 --warmup 3 > $R/$O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $R/$O/pmc_sq -o s -- $B --steps 5 --warmup 2 --no-overlap > $R/$O/pmc_sq.log 2>&1
cd $R
db() { find $O/$1 -name '*.db' | head -1; }
python profiles/summarize_rocpd.py $(db prof_alone) > $O/kernel_stats.csv 2>&1
python profiles/pmc_summary.py $(db pmc_fetch) $(db pmc_write) 32 640 480 "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- $B --steps 6 --warmup 3" borrow > $O/pmc_summary.json 2> $O/pmc_summary.err
python profiles/pmc_by_kernel.py $(db pmc_sq) k_ > $O/pmc_wait_and_instructions.txt 2>&1
head -12 $O/kernel_stats.csv | cut -c1-120
grep -A4 '"k_track' $O/pmc_summary.json | head -8
find $O -name '*.db' -delete
