#!/bin/bash
# round 5, GPU call 23: final lines with the depth half of pyrDown on the auxiliary stream -- the whole -m gpu suite, then the three bench lines again
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c23; mkdir -p $O
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error|Error" $O/pytest_gpu.log | head -5
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; grep WARNING $O/bench_default.err
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; grep WARNING $O/bench_driver_args.err
timeout 400 python bench.py --width 1280 --height 960 --levels 5 --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 30 --warmup 5 > $O/bench_1280x960x5.json 2> $O/bench_1280.err
python - <<'PY'
import json
for n in ("bench_default", "bench_driver_args", "bench_1280x960x5"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5c23/%s.json" % n) if l.startswith("{")][-1])
        r = d["roofline"]
        print(n, "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "frac", round(r["frac"], 3), "alone", round(r["frac_alone"], 3), "step", round(r["step"]["frac"], 3), "traffic", r["traffic"], r["traffic_commit"], "gate", d["resident_gate"]["timeouts"], "coll", d["collective"]["steps_per_collective"])
        ss = d.get("single_stream") or {}
        if ss: print("   single_stream", round(ss["frames_per_s"]), [round(x) for x in ss["frames_per_s_runs"]])
    except Exception as e:
        print(n, "FAILED", e)
PY
