#!/bin/bash
# round 5, GPU call 13: the two configurations call 12 fed from a cache file of another geometry / size (the bench now refuses that)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c12; mkdir -p $O
C="--cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 30 --warmup 5"
for spec in "p64:--pairs 64" "hd:--width 1920 --height 1080 --levels 4 --pairs 8 --steps 20"; do
  n=${spec%%:*}; a=${spec#*:}
  timeout 200 python bench.py $C $a > $O/bench_$n.json 2> $O/bench_$n.err
done
python - <<'PY'
import json, glob
for f in ("gpurun_out/r5c12/bench_p64.json", "gpurun_out/r5c12/bench_hd.json"):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["config"]["workload"][:70], "|", round(d["value"]), "frames/s", round(d["ms_per_step"], 4), "ms/step", d["stages_ms"], "gate", d["resident_gate"]["timeouts"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-400:])
PY
