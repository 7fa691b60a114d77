#!/bin/bash
# round 4, GPU call 14: fillInEdges with a launch per level on large images: parity at the large sizes, 1280x960x5 bench
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c14; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "1280 or unusual or other_level or edge_cases or large_level" ) > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log
timeout 200 python tests/tools/soak_gpu_parity.py 16 5 2>&1 | grep -v amdgpu.ids | tail -17 | tee $O/build_soak.txt
timeout 200 python bench.py --width 1280 --height 960 --levels 5 --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 30 --warmup 5 > $O/bench_1280x960x5.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/c14/bench_1280x960x5.json") if l.startswith("{")][-1])
print("1280x960x5", round(d["value"]), round(d["ms_per_step"], 4), d["stages_ms"])
PY
