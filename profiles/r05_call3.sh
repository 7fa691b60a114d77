#!/bin/bash
# round 5, GPU call 3: the default bench line collapsed to 134 frames/s in call 1 (every resident gate timed out) while the A/B runs
# of the same loop without the sequential-stream leg ran at 99 k: which leg of the full run breaks the census?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c3; mkdir -p $O
B="python bench.py --cpu-baseline off --skip-host-buffers --steps 12 --warmup 4 --input-cache /tmp/revo_c3_inputs"
run() { echo "== $1"; shift; ( "$@" ) > $O/tmp.json 2> $O/tmp.err; grep WARNING $O/tmp.err; python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r5c3/tmp.json") if l.startswith("{")][-1])
    print("value", round(d["value"]), "ms", round(d["ms_per_step"], 3), "gate", d.get("resident_gate"), "ss", (d.get("single_stream") or {}).get("frames_per_s"), "pipe", {k: d["config"]["pipeline"].get(k) for k in ("distinct_hw_queues", "streams_replaced")})
except Exception as e:
    print("FAILED", e); print(open("gpurun_out/r5c3/tmp.err").read()[-600:])
PY
}
run "default legs (single stream 20 frames)" timeout 300 $B --single-stream-frames 20 --single-stream-runs 2
run "no single stream" timeout 300 $B --single-stream-frames 0
run "single stream, REVO_DIRECT_H2D=0" timeout 300 env REVO_DIRECT_H2D=0 $B --single-stream-frames 20 --single-stream-runs 2
run "single stream, no collective" timeout 300 $B --single-stream-frames 20 --single-stream-runs 2 --no-collective
run "single stream, shape bench" timeout 300 $B --single-stream-frames 20 --single-stream-runs 2 --shape bench
run "single stream, no probe" timeout 300 env REVO_PIPE_PROBE=0 $B --single-stream-frames 20 --single-stream-runs 2
