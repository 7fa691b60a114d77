#!/bin/bash
# round 4, GPU call 11: the whole -m gpu suite at HEAD (new: the three-stage pipeline test) + smoke + repeatability of the bench lines
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c11; mkdir -p $O
( time timeout 500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error|soak" $O/pytest_gpu.log
timeout 120 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -2
for i in 1 2; do
  timeout 300 python bench.py > $O/bench_default_$i.json 2> $O/bench_default_$i.err
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args_$i.json 2> $O/bench_driver_args_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c11/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 4), "single", round(d["value_single_batch_in_flight"]), "two", round(d["value_two_batches"]),
              "ss", [round(x) for x in d["single_stream"]["frames_per_s_runs"]], "h2d", [round(x, 1) for x in d["host_buffers"]["u16"]["pcie_gbs_runs"]], [round(x, 1) for x in d["host_buffers"]["f32"]["pcie_gbs_runs"]])
    except Exception as e:
        print(f, "FAILED", e)
PY
