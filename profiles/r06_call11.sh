#!/bin/bash
# round 6, GPU call 11: staged edge depths (REVO_STAGE_EDGE_DEPTHS=1 against 0): parity tests, bench A/B, PMC traffic of the edge-list kernels
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r6c11; mkdir -p $O
R=$GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_hostbatch.py tests/test_gpu_tracker2.py tests/test_gpu_zz_deferred_edt.py -m gpu -x -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for st in 1 0 1 0; do
  REVO_STAGE_EDGE_DEPTHS=$st timeout 600 python bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --steps 80 --warmup 8 > $O/bench_st$st.json 2> $O/bench_st$st.err
  python - $st <<'PY'
import json,sys
c=sys.argv[1]
try:
    d = json.loads([l for l in open("gpurun_out/r6c11/bench_st%s.json" % c) if l.startswith("{")][-1])
    k = {x["kernel"]: round(x["us_alone"], 1) for x in d["roofline"]["kernels"] if any(t in x["kernel"] for t in ("pyrdown_depth", "tile", "pts", "edge_prefix"))}
    print("stage", c, "value", round(d["value"]), "ms", round(d["ms_per_step"], 4), "build_ms", round(d["stages_ms"]["pyramids_and_keyframes"], 4), k)
except Exception as e:
    print(c, "FAILED", e); print(open("gpurun_out/r6c11/bench_st%s.err" % c).read()[-1200:])
PY
done 2>&1 | tee $O/ab_stage_edge_depths.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --no-collective --render-procs 1 --input-cache /tmp/revo_r6_inputs"
timeout 300 $B --steps 6 --warmup 2 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc_fetch -o f -- $B --steps 6 --warmup 3 > $R/$O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/pmc_write -o w -- $B --steps 6 --warmup 3 > $R/$O/pmc_write.log 2>&1
cd $R
db() { find $O/$1 -name '*.db' | head -1; }
python profiles/pmc_summary.py $(db pmc_fetch) $(db pmc_write) 32 640 480 "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- $B --steps 6 --warmup 3" borrow > $O/pmc_summary.json 2> $O/pmc_summary.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6c11/pmc_summary.json'))
for k,v in d.items():
    if isinstance(v,dict) and 'hbm_bytes_per_launch' in v and k.startswith('k_'): print(k, round(v['hbm_bytes_per_launch']/1e6,1),'MB')
PY
find $O -name '*.db' -delete
