#!/bin/bash
# Builds one variant library per experiment branch into profiles/build/librevo_hip_var_<name>.so without
# touching the working tree: every branch is checked out into a throw-away git worktree under /tmp and compiled from there.
# Run it HERE (the GPU box has no .git); the libraries travel with the gpurun snapshot.  Then on the box, per variant:
#   REVO_HIP_SO=profiles/build/librevo_hip_var_<name>.so python -m pytest tests -m gpu -x -q     (parity first)
#   python profiles/ab_bench.py base= <name>=profiles/build/librevo_hip_var_<name>.so            (then the A/B)
# A branch whose host code has fallen behind main must be rebased first (the variant library is the WHOLE library).
set -e
cd /root/repo
for spec in "edtlean:exp/edt-rows-lean" "reuse:exp/retry-patch-reuse" "nms12:exp/nms-12rows"; do
  name=${spec%%:*}; branch=${spec#*:}
  wt=/tmp/revo_wt_$name
  git worktree remove --force "$wt" 2>/dev/null || true
  git worktree add -q --detach "$wt" "$branch"
  base=$(git merge-base main "$branch")
  if ! git diff --quiet "$base" main -- revo_amd/csrc include; then
    echo "NOTE: main has changed revo_amd/csrc or include/ since $branch left it: rebase the branch before trusting the A/B"
  fi
  SRC_DIR=$wt/revo_amd/csrc SUFFIX=_$name SHOW="${SHOW:-k_edt|k_track|k_canny}" bash profiles/build_var.sh | tail -4
  git worktree remove --force "$wt"
done
ls -la profiles/build/
