#!/bin/bash
# round 6, GPU call 5: VO tests (640x480 x 120 frames, decoder pool, bench --tum-dir), bench --coll native vs torch, lowered tracker caps
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c5; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_vo.py -m gpu -x -q -s ) > $O/pytest_vo.log 2>&1; grep -E "passed|failed|error|Error|640x480x4|sequential stream from|ATE" $O/pytest_vo.log | head -12
( time timeout 900 python -m pytest tests/test_gpu_tracker2.py tests/test_gpu_variants.py tests/test_gpu_parity.py -m gpu -x -q ) > $O/pytest_tracker.log 2>&1; tail -3 $O/pytest_tracker.log
for coll in native torch native torch; do
  timeout 500 python bench.py --coll $coll --cpu-baseline off --skip-host-buffers --single-stream-frames 0 --steps 80 --warmup 8 > $O/bench_$coll.json 2> $O/bench_$coll.err
  python - $coll <<'PY'
import json,sys
c=sys.argv[1]
try:
    d = json.loads([l for l in open("gpurun_out/r6c5/bench_%s.json" % c) if l.startswith("{")][-1])
    print(c, "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), d["collective"]["backend"][:60], d["collective"]["us_per_all_gather_alone"])
except Exception as e:
    print(c, "FAILED", e); print(open("gpurun_out/r6c5/bench_%s.err" % c).read()[-1500:])
PY
done
