#!/bin/bash
# round 5, GPU call 24: the driver's own sequence on the final commit -- pytest -m gpu, smoke(), bench with the driver's arguments
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c24; mkdir -p $O
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | head -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -1
timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; grep WARNING $O/bench_driver_args.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5c24/bench_driver_args.json") if l.startswith("{")][-1])
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "traffic", d["roofline"]["traffic"], "gate", d["resident_gate"], "ss", round(d["single_stream"]["frames_per_s"]))
PY
