cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c2; mkdir -p $O
for v in "pageable thread" "pinned thread" "pinned nothread"; do echo "== $v"; timeout 200 python profiles/census_probe.py $v 2>&1 | grep -v "^\[W\|amdgpu.ids"; done | tee $O/census_probe.txt
