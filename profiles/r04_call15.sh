#!/bin/bash
# round 4, GPU call 15: closing validation at HEAD -- the whole -m gpu suite, smoke, the default bench line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c15; mkdir -p $O
( time timeout 400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error|soak" $O/pytest_gpu.log
timeout 100 python __graft_entry__.py smoke 2>&1 | grep "smoke"
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/c15/bench_driver_args.json") if l.startswith("{")][-1])
print("driver args", round(d["value"]), round(d["ms_per_step"], 4), d["roofline"]["frac"], d["collective"]["ranks_seen"])
PY
