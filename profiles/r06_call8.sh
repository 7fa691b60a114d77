#!/bin/bash
# round 6, GPU call 8: the MIXED hysteresis (heavy level-0 frames through bands inside the single-workgroup launch): parity tests, stage times, bench A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c8; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for hv in 4000 0 2500 6000; do
  REVO_HYST_HEAVY_RUNS=$hv timeout 600 python bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 80 --warmup 8 > $O/bench_hv$hv.json 2> $O/bench_hv$hv.err
  python - $hv <<'PY'
import json,sys
c=sys.argv[1]
try:
    d = json.loads([l for l in open("gpurun_out/r6c8/bench_hv%s.json" % c) if l.startswith("{")][-1])
    k = {x["name"]: round(x["us"], 1) for x in d["roofline"]["kernels"] if "hyst" in x["name"] or "nms" in x["name"]}
    print("heavy_runs", c, "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "build_ms", round(d["stages_ms"]["pyramids_and_keyframes"], 4), k)
except Exception as e:
    print(c, "FAILED", e); print(open("gpurun_out/r6c8/bench_hv%s.err" % c).read()[-1200:])
PY
done 2>&1 | tee $O/ab_mixed_hyst.txt
