#!/bin/bash
# round 6, GPU call 9: kernel trace of the build alone with the mixed hysteresis on (4000) and off (0): where do its 98 us go?
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=$PWD/gpurun_out/r6c9; mkdir -p $O
export TMPDIR=/tmp
for hv in 4000 0; do
  ( cd /tmp && REVO_HYST_HEAVY_RUNS=$hv timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_hv$hv -o hv$hv -- python $GRAFT_REPO_ROOT/profiles/build_only.py > $O/prof_hv$hv.log 2>&1 )
  f=$(find $O/prof_hv$hv -name "*kernel_stats.csv" | head -1)
  echo "== REVO_HYST_HEAVY_RUNS=$hv ($f)"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r.get("Name") or r.get("KernelName")
    if any(k in n for k in ("hyst","nms","fill")):
        print("%-60s calls %4s avg %8.1f us min %8.1f max %8.1f" % (n[:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done 2>&1 | tee $O/hyst_kernel_times.txt
