#!/bin/bash
# round 6, GPU call 3: + opaque lane ids in the solver / reduction sections, cluster_poll specialised on the cluster size (SGPR spills 174 -> 71)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c3; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_tracker2.py -m gpu -x -q ) > $O/pytest_tracker.log 2>&1; grep -E "passed|failed|error" $O/pytest_tracker.log | head -5
REVO_HIP_SO=profiles/build/librevo_hip_prof.so timeout 300 python profiles/batch_phases.py > $O/batch_phases.txt 2>&1; tail -12 $O/batch_phases.txt
REVO_HIP_SO=profiles/build/librevo_hip_prof.so PH_KSPEC=2244 PH_CLUSTER=16 timeout 300 python profiles/single_pair_phases.py > $O/single_pair_phases.txt 2>&1; tail -3 $O/single_pair_phases.txt
timeout 300 python profiles/batch_phases.py 2>&1 | grep "k_track alone" | tee $O/track_alone_production.txt
PH_KSPEC=2244 PH_CLUSTER=16 timeout 300 python profiles/single_pair_phases.py 2>&1 | tail -1 | tee $O/single_pair_production.txt
timeout 500 python bench.py --cpu-baseline off --skip-host-buffers > $O/bench_short.json 2> $O/bench_short.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6c3/bench_short.json") if l.startswith("{")][-1])
r = d["roofline"]
print("value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "frac", round(r["frac"], 3), "alone", round(r["frac_alone"], 3), "kernel_ms", r.get("kernel_ms"), r.get("kernel_ms_alone"))
ss = d.get("single_stream") or {}
if ss: print("   single_stream", round(ss["frames_per_s"]), [round(x) for x in ss["frames_per_s_runs"]], ss.get("trajectory_rmse_gpu_vs_oracle_m"))
PY
