#!/bin/bash
# round 5, GPU call 1: the whole -m gpu suite on the round's host-side changes (pipeline handle, per-context knobs, depth-agnostic
# tracker chain, ADVICE r04 ordering, direct H2D of page-locked frames), then the library-owned pipeline against round 4's
# bench-owned loop (interleaved A/B) and with the stream probe switched off, then the default bench line.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c1; mkdir -p $O
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error|soak|Error" $O/pytest_gpu.log | head -20
timeout 900 python profiles/ab_bench.py --runs 2 lib= 'bench=@--shape bench' 'noprobe=REVO_PIPE_PROBE=0' 2>&1 | tee $O/ab_lib_vs_bench.txt
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r5c1/bench_default.json") if l.startswith("{")][-1])
    print("value", round(d["value"]), "ms", round(d["ms_per_step"], 4), d["stages_ms"], "single", d.get("value_single_batch_in_flight"), "two", d.get("value_two_batches"))
    print("pipeline", d["config"]["pipeline"])
    print("roofline frac", d["roofline"]["frac"], "alone", d["roofline"]["frac_alone"], "step", d["roofline"]["step"])
    for k in d["roofline"]["kernels"]:
        print("  ", k)
    print("single_stream", {k: v for k, v in (d.get("single_stream") or {}).items() if k in ("frames_per_s", "frames_per_s_runs", "speedup_vs_cpu_oracle_2core_pipelined")})
    print("host", {k: (v.get("value_incl_h2d_runs"), v.get("warmup_groups_of_3_jobs_s")) for k, v in (d.get("host_buffers") or {}).items() if isinstance(v, dict)})
    print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu_baseline"))
except Exception as e:
    print("bench_default FAILED", e)
PY
