#!/bin/bash
# round 4, GPU call 8: the round's final evidence on the shipped default -- bench lines (default / the driver's arguments /
# 1280x960x5), the whole -m gpu suite, kernel trace + PMC passes of the pipelined step.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c8; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error|soak" $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json | head -c 300; echo
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
timeout 300 python bench.py --width 1280 --height 960 --levels 5 --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 30 --warmup 5 > $O/bench_1280x960x5.json 2> $O/bench_1280.err
python - <<'PY'
import json
for n in ("bench_default", "bench_driver_args", "bench_1280x960x5"):
    try:
        d = json.loads([l for l in open("gpurun_out/c8/%s.json" % n) if l.startswith("{")][-1])
        print(n, round(d["value"]), round(d["ms_per_step"], 4), d["stages_ms"], "single", d.get("value_single_batch_in_flight"), "two", d.get("value_two_batches"),
              "frac", round(d["roofline"]["frac"], 3), "ss", (d.get("single_stream") or {}).get("frames_per_s"))
    except Exception as e:
        print(n, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --no-collective --render-procs 1 --input-cache /tmp/revo_c8_inputs"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- $B --steps 24 --warmup 4 > $R/$O/bench_profiled_overlapped.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- $B --steps 6 --warmup 3 > $R/$O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- $B --steps 6 --warmup 3 > $R/$O/pmc_write.log 2>&1
cd $R
db() { find $O/$1 -name '*.db' | head -1; }
python profiles/summarize_rocpd.py $(db prof) > $O/kernel_stats_overlapped.csv 2>&1
python profiles/stream_timeline.py $(db prof) 12 > $O/step_timeline.txt 2>&1
python profiles/stream_timeline.py $(db prof) 16 > $O/step_timeline_b.txt 2>&1
python profiles/track_overlap.py $(db prof) > $O/track_overlap.txt 2>&1
python profiles/overlap_slowdown.py $(db prof) > $O/overlap.txt 2>&1
python profiles/pmc_summary.py $(db pmc_fetch) $(db pmc_write) 32 640 480 "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- $B --steps 6 --warmup 3" borrow > $O/pmc_summary.json 2> $O/pmc_summary.err
cat $O/step_timeline.txt
find $O -name '*.db' -delete
