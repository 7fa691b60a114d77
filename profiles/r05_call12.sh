#!/bin/bash
# round 5, GPU call 12: the hand-run soaks on the final sources (256 tracker pairs vs the oracle; 32 randomised build scenes incl.
# 1280x1024 / 1920x1080), the N > 1 gather schedule on RCCL at world size 1 (--gather-every 2 and 3), other configurations for the record
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c12; mkdir -p $O
timeout 400 python tests/tools/soak_gpu_tracker.py 256 1000 2>&1 | grep -v amdgpu.ids | tee $O/parity_soak.txt | tail -6
timeout 300 python tests/tools/soak_gpu_parity.py 32 2>&1 | grep -v amdgpu.ids | tail -34 > $O/build_soak.txt; tail -3 $O/build_soak.txt
C="--cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 30 --warmup 5 --input-cache /tmp/revo_c12"
for spec in "ge2:--gather-every 2" "ge3:--gather-every 3 --steps 31" "lv3:--levels 3" "p16:--pairs 16" "p64:--pairs 64" "hd:--width 1920 --height 1080 --levels 4 --pairs 8 --steps 20"; do
  n=${spec%%:*}; a=${spec#*:}
  timeout 150 python bench.py $C $a > $O/bench_$n.json 2> $O/bench_$n.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5c12/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], d["config"]["workload"][:70], "|", round(d["value"]), "frames/s", round(d["ms_per_step"], 4), "ms/step", d["stages_ms"], "coll", d["collective"]["steps_per_collective"], "gate", d["resident_gate"]["timeouts"])
    except Exception as e:
        print(f, "FAILED", e, open(f.replace(".json", ".err")).read()[-400:])
PY
