#!/bin/bash
# round 4, GPU call 9: where a frame of the sequential stream spends its host time; the default bench line again (single-stream runs
# with the collector kept out of the timed runs)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c9; mkdir -p $O
timeout 120 python profiles/single_stream_profile.py 60 4 2>&1 | grep -v amdgpu.ids | tee $O/single_stream_profile.txt
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/c9/bench_default.json") if l.startswith("{")][-1])
print(round(d["value"]), round(d["ms_per_step"], 4), d["single_stream"]["frames_per_s_runs"], d["host_buffers"]["u16"]["pcie_gbs_runs"], d["host_buffers"]["f32"]["pcie_gbs_runs"])
PY
