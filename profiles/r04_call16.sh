#!/bin/bash
# round 4, GPU call 16: other configurations for the record (3 levels, 16 / 64 pairs per step, 1920x1080x4)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c16; mkdir -p $O
C="--cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 30 --warmup 5"
for spec in "lv3:--levels 3" "p16:--pairs 16" "p64:--pairs 64" "hd:--width 1920 --height 1080 --levels 4 --pairs 8 --steps 20"; do
  n=${spec%%:*}; a=${spec#*:}
  timeout 150 python bench.py $C $a > $O/bench_$n.json 2> $O/bench_$n.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c16/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["config"]["workload"][:60], "|", round(d["value"]), "frames/s", round(d["ms_per_step"], 4), "ms/step", d["stages_ms"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-300:])
PY
