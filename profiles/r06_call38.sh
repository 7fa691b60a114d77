#!/bin/bash
# round 6, GPU calls 19, 32, 38 (each on the then-final tree): the driver's sequence (full GPU suite, smoke, bench with the driver's
# arguments and with the defaults)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c38; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -3 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; tail -c 600 $O/bench_driver_args.err
python - <<'PY'
import json
for f in ("gpurun_out/r6c38/bench_driver_args.json",):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "steps", "warmup")}, d["roofline"], d["cpu_baseline"], d["single_stream"]["frames_per_s"])
PY
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6c38/bench_default.json") if l.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["roofline"]["frac"], d["single_stream"]["frames_per_s"], [round(x) for x in d["single_stream"]["frames_per_s_runs"]])
PY
