#!/bin/bash
# round 4, GPU call 18: twelve full-length runs of the sequential stream in one process: how often is a run slow?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out/c18
timeout 100 python bench.py --cpu-baseline off --skip-host-buffers --single-stream-runs 12 --steps 10 --warmup 2 --no-collective > gpurun_out/c18/b.json 2> gpurun_out/c18/b.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/c18/b.json") if l.startswith("{")][-1])
print("single-stream runs (frames/s):", [round(x) for x in d["single_stream"]["frames_per_s_runs"]])
PY
