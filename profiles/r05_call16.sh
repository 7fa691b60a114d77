#!/bin/bash
# round 5, GPU call 16: flakiness check before the round ends -- the -m gpu suite three times over, smoke()
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c16; mkdir -p $O
for i in 1 2 3; do ( time timeout 400 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/pytest_gpu_$i.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_$i.log | head -3; done
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2
