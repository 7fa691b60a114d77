#!/bin/bash
# round 5, GPU call 14: the round's final bench lines with the final default (two steps' records per collective at every N)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c14; mkdir -p $O
timeout 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; grep WARNING $O/bench_default.err
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; grep WARNING $O/bench_driver_args.err
timeout 400 python bench.py --width 1280 --height 960 --levels 5 --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 30 --warmup 5 > $O/bench_1280x960x5.json 2> $O/bench_1280.err
timeout 200 python bench.py --shape bench --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 13 --warmup 3 > $O/bench_shape_bench.json 2> $O/bench_shape_bench.err
python - <<'PY'
import json
for n in ("bench_default", "bench_driver_args", "bench_1280x960x5", "bench_shape_bench"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5c14/%s.json" % n) if l.startswith("{")][-1])
        r = d["roofline"]
        print(n, "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), d["stages_ms"], "single", round(d.get("value_single_batch_in_flight") or 0), "two", round(d.get("value_two_batches") or 0),
              "frac", round(r["frac"], 3), "alone", round(r["frac_alone"], 3), "step", round(r["step"]["frac"], 3), "traffic", r["traffic"], "gate", d["resident_gate"], "coll", d["collective"]["steps_per_collective"], d["config"]["pipeline"].get("steps_submitted"))
        ss = d.get("single_stream") or {}
        if ss: print("   single_stream", round(ss["frames_per_s"]), [round(x) for x in ss["frames_per_s_runs"]], ss.get("speedup_vs_cpu_oracle_2core_pipelined"))
        hb = d.get("host_buffers") or {}
        for k, v in hb.items():
            if isinstance(v, dict): print("   host", k, round(v["value_incl_h2d"]), [round(x) for x in v["value_incl_h2d_runs"]], "first", round(v["value_incl_h2d_first_repetition"]), round(v["pcie_gbs"], 1), "GB/s")
        if "cpu_baseline" in d: print("   cpu", round(d["cpu_baseline"]["value"], 1), "x", round(d["speedup_vs_cpu_baseline"]), "1core", d.get("cpu_baseline_1core", {}).get("value"), "all", d.get("cpu_baseline_all_cores", {}).get("value"))
    except Exception as e:
        print(n, "FAILED", e)
PY
