#!/bin/bash
# round 4, GPU call 13: 1280x960x5 (BASELINE configs[3]): kernel times alone, hysteresis band size
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/c13; mkdir -p $O
R=$GRAFT_REPO_ROOT
A="--width 1280 --height 960 --levels 5 --steps 30 --warmup 5"
timeout 500 python profiles/ab_bench.py --runs 1 --args "$A" base= 'bw1200=REVO_HYST_BAND_WORDS=1200' 'bw4800=REVO_HYST_BAND_WORDS=4800' 'b3=@--buffers 3' 'd1=REVO_DEFER=1' 2>&1 | tee $O/ab_1280x960.txt
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-baseline off --single-stream-frames 0 --skip-host-buffers --no-collective --render-procs 1 --input-cache /tmp/revo_c13_inputs $A"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_alone -o a -- $B --no-overlap --steps 8 --warmup 2 > $R/$O/bench_prof_alone.log 2>&1
cd $R
python profiles/summarize_rocpd.py $(find $O/prof_alone -name '*.db' | head -1) > $O/kernel_stats_1280x960x5.csv 2>&1
cut -c1-70 $O/kernel_stats_1280x960x5.csv | head -3; awk -F, 'NR>1{n=split($1,a,"::"); printf "%-40s calls %s avg %s\n", substr(a[n],1,40), $(NF-7), $(NF-5)}' $O/kernel_stats_1280x960x5.csv | head -20
find $O -name '*.db' -delete
