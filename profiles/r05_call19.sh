#!/bin/bash
# round 5, GPU call 19: steps per collective at N = 1 (the gather sits in the after-grid slot; ProcessGroupNCCL runs it on a stream of
# its own behind an event, i.e. a fifth active stream for its duration): 1 / 2 / 4 / 8 / one at the end
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c19; mkdir -p $O
timeout 1200 python profiles/ab_bench.py --runs 2 'ge1=@--gather-every 1' 'ge2=@--gather-every 2' 'ge4=@--gather-every 4' 'ge8=@--gather-every 8' 'ge88=@--gather-every 88' 'nocoll=@--no-collective' 2>&1 | tee $O/ab_gather_every.txt
