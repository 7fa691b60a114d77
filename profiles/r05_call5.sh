#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c5; mkdir -p $O
for v in "sync" "nosync"; do echo "== $v"; timeout 90 python profiles/census_probe2.py $v 2>&1 | grep -v "^\[W\|amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl"; done | tee $O/census_probe2.txt
B="python bench.py --cpu-baseline off --skip-host-buffers --steps 12 --warmup 4 --input-cache /tmp/revo_c5_inputs --single-stream-frames 20 --single-stream-runs 2"
for i in 1 2; do timeout 200 $B > $O/b$i.json 2> $O/b$i.err; grep WARNING $O/b$i.err; python -c "
import json
d=json.loads([l for l in open('$O/b$i.json') if l.startswith('{')][-1]); print('bench', round(d['value']), d['resident_gate'])"; done
