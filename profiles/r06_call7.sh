#!/bin/bash
# round 6, GPU call 7: decode-pool rates on the GPU box's host (no GPU path), and the size of the merged H2D copies of the host-buffer batches
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c7; mkdir -p $O
timeout 600 python profiles/decode_rates.py 2>&1 | tee $O/decode_rates.txt | tail -16
for mb in 2 8 24 64 1024; do
  REVO_H2D_MAX_RUN_MB=$mb timeout 600 python bench.py --cpu-baseline off --single-stream-frames 0 --steps 20 --warmup 5 > $O/bench_run$mb.json 2> $O/bench_run$mb.err
  python - $mb <<'PY'
import json,sys
c=sys.argv[1]
try:
    d = json.loads([l for l in open("gpurun_out/r6c7/bench_run%s.json" % c) if l.startswith("{")][-1])
    h = d["host_buffers"]
    print("max run MB", c, {k: (round(h[k]["value_incl_h2d"]), round(h[k]["pcie_gbs"], 1)) for k in ("u16", "f32")})
except Exception as e:
    print(c, "FAILED", e)
PY
done 2>&1 | tee $O/h2d_run_sizes.txt
