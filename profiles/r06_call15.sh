#!/bin/bash
# round 6, GPU call 15 (the final sources; call 10 was the same script before the staged-depth / mixed-hysteresis experiments): evidence on the round's sources: the whole -m gpu suite, the three bench lines, rocprofv3 kernel stats alone and in the
# pipelined step, step timeline, PMC passes (FETCH_SIZE / WRITE_SIZE separately; wait + instruction counters), the sequential stream's
# device timeline, the 256-pair tracker soak
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c15; mkdir -p $O
R=$GRAFT_REPO_ROOT
export REVO_COMMIT=$(cat profiles/.r06_commit 2>/dev/null || echo unknown)
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error|Error" $O/pytest_gpu.log | head -5
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; grep WARNING $O/bench_default.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; grep WARNING $O/bench_driver_args.err
timeout 400 python bench.py --width 1280 --height 960 --levels 5 --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 30 --warmup 5 > $O/bench_1280x960x5.json 2> $O/bench_1280.err
timeout 400 python bench.py --coll native --cpu-baseline off --single-stream-frames 0 --skip-host-buffers > $O/bench_coll_native.json 2> $O/bench_native.err
python - <<'PY'
import json
for n in ("bench_default", "bench_driver_args", "bench_1280x960x5", "bench_coll_native"):
    try:
        d = json.loads([l for l in open("gpurun_out/r6c15/%s.json" % n) if l.startswith("{")][-1])
        r = d["roofline"]
        print(n, "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "frac", round(r["frac"], 3), "alone", round(r["frac_alone"], 3), "step", round(r["step"]["frac"], 3), "traffic", r["traffic"], "gate", d["resident_gate"]["timeouts"], "coll", d["collective"]["steps_per_collective"])
        ss = d.get("single_stream") or {}
        if ss: print("   single_stream", round(ss["frames_per_s"]), [round(x) for x in ss["frames_per_s_runs"]], ss.get("speedup_vs_cpu_oracle_2core_pipelined"))
        hb = d.get("host_buffers") or {}
        if hb: print("   host", {k: (round(hb[k]["value_incl_h2d"]), round(hb[k]["pcie_gbs"], 1)) for k in ("u16", "f32")})
        cb = d.get("cpu_baseline")
        if cb: print("   cpu", cb.get("value"), cb.get("cores"))
    except Exception as e:
        print(n, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --no-collective --render-procs 1 --input-cache /tmp/revo_r6_inputs"
timeout 300 $B --steps 6 --warmup 2 > /dev/null 2>&1   # renders the inputs once (the profiler deadlocks on the render pool)
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_alone -o p -- $B --steps 10 --warmup 3 --no-overlap > $R/$O/bench_profiled_alone.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- $B --steps 24 --warmup 4 > $R/$O/bench_profiled_overlapped.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- $B --steps 6 --warmup 3 > $R/$O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- $B --steps 6 --warmup 3 > $R/$O/pmc_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $R/$O/pmc_wait -o q -- $B --steps 5 --warmup 2 --no-overlap > $R/$O/pmc_wait.log 2>&1
( cd $R && timeout 300 rocprofv3 --kernel-trace -d $R/$O/prof_seq -o s -- python profiles/single_stream_profile.py 120 4 > $R/$O/single_stream_profile.txt 2>&1 )
cd $R
db() { find $O/$1 -name '*.db' | head -1; }
python profiles/summarize_rocpd.py $(db prof_alone) > $O/kernel_stats.csv 2>&1
python profiles/summarize_rocpd.py $(db prof) > $O/kernel_stats_overlapped.csv 2>&1
python profiles/stream_timeline.py $(db prof) 12 > $O/step_timeline.txt 2>&1
python profiles/track_overlap.py $(db prof) > $O/track_overlap.txt 2>&1
python profiles/overlap_slowdown.py $(db prof) > $O/overlap.txt 2>&1
python profiles/pmc_summary.py $(db pmc_fetch) $(db pmc_write) 32 640 480 "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- $B --steps 6 --warmup 3" borrow > $O/pmc_summary.json 2> $O/pmc_summary.err
python profiles/pmc_by_kernel.py $(db pmc_wait) k_ > $O/pmc_wait_and_instructions.txt 2>&1
python profiles/seq_timeline.py $(db prof_seq) 12 3 > $O/single_stream_timeline.txt 2>&1
python profiles/stream_busy.py $(db prof) > $O/stream_busy.txt 2>&1
head -24 $O/step_timeline.txt
head -14 $O/kernel_stats.csv | cut -c1-150
head -12 $O/single_stream_timeline.txt; grep -v amdgpu $O/single_stream_profile.txt | tail -4
find $O -name '*.db' -delete
timeout 500 python tests/tools/soak_gpu_tracker.py 256 1000 2>&1 | grep -v amdgpu.ids | tee $O/parity_soak.txt | tail -6
