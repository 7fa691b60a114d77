#!/bin/bash
# round 4, GPU call 10: order of the deferred kernels (edge lists / EDT) on their stream
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c10; mkdir -p $O
timeout 400 python profiles/ab_bench.py --runs 2 base= 'edtfirst=REVO_AUX_ORDER=1' 'edtfirst_d3=REVO_AUX_ORDER=1,REVO_DEFER=3' 2>&1 | tee $O/ab_aux_order.txt
