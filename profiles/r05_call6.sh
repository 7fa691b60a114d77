#!/bin/bash
# round 5, GPU call 6: k_hyst phase cycles per frame (profiling build), the whole -m gpu suite on the fixed head, the bench line
# with the driver's arguments
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c6; mkdir -p $O
REVO_HIP_SO=profiles/build/librevo_hip_var_hp.so timeout 120 python profiles/hyst_profile.py 2>&1 | grep "^hyst f=" | sort -t= -k2 -n > $O/hyst_profile.txt; wc -l $O/hyst_profile.txt; head -3 $O/hyst_profile.txt
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error|soak|Error" $O/pytest_gpu.log | head
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; grep WARNING $O/bench_driver_args.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r5c6/bench_driver_args.json") if l.startswith("{")][-1])
    print("value", round(d["value"]), "ms", round(d["ms_per_step"], 4), d["stages_ms"], "single", round(d.get("value_single_batch_in_flight") or 0), "two", round(d.get("value_two_batches") or 0), "gate", d["resident_gate"])
    print("roofline frac", round(d["roofline"]["frac"], 3), "alone", round(d["roofline"]["frac_alone"], 3), "step frac", round(d["roofline"]["step"]["frac"], 3), "kernel_ms", d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_alone"])
    for k in d["roofline"]["kernels"]:
        print("  %-14s %8.1f us  frac %s" % (k["kernel"], k["us_alone"], None if k.get("frac") is None else round(k["frac"], 3)))
    ss = d.get("single_stream") or {}
    print("single_stream", ss.get("frames_per_s"), ss.get("frames_per_s_runs"), ss.get("speedup_vs_cpu_oracle_2core_pipelined"))
    print("host", {k: (v.get("value_incl_h2d_runs"), v.get("warmup_groups_of_3_jobs_s")) for k, v in (d.get("host_buffers") or {}).items() if isinstance(v, dict)})
    print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu_baseline"))
except Exception as e:
    print("bench FAILED", e); print(open("gpurun_out/r5c6/bench_driver_args.err").read()[-800:])
PY
