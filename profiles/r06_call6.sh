#!/bin/bash
# round 6, GPU call 6: host-buffer slabs (one H2D copy per plane type and job) vs one buffer per frame; decoder pool test; hostbatch tests
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c6; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_hostbatch.py "tests/test_gpu_vo.py::test_decoder_pool_keeps_the_sequential_stream_fed" tests/test_gpu_vo.py::test_bench_runs_a_tum_layout_folder -m gpu -x -q -s ) > $O/pytest.log 2>&1; grep -E "passed|failed|rror|sequential stream from" $O/pytest.log | head
for slabs in 1 0; do
  REVO_BENCH_HOST_SLABS=$slabs timeout 600 python bench.py --cpu-baseline off --single-stream-frames 0 --steps 20 --warmup 5 > $O/bench_slabs$slabs.json 2> $O/bench_slabs$slabs.err
  python - $slabs <<'PY'
import json,sys
c=sys.argv[1]
try:
    d = json.loads([l for l in open("gpurun_out/r6c6/bench_slabs%s.json" % c) if l.startswith("{")][-1])
    h = d["host_buffers"]
    print("slabs", c, "value", round(d["value"]), {k: (round(h[k]["value_incl_h2d"]), round(h[k]["pcie_gbs"], 1), [round(x,1) for x in h[k]["pcie_gbs_runs"]], round(h[k]["value_incl_h2d_first_repetition"])) for k in ("u16", "f32")})
except Exception as e:
    print(c, "FAILED", e); print(open("gpurun_out/r6c6/bench_slabs%s.err" % c).read()[-1500:])
PY
done
