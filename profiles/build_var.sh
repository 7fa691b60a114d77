#!/bin/bash
# variant build of the library for A/B runs -> profiles/build/librevo_hip_var${SUFFIX}.so   (EXTRA_DEFS="-D...";
# SRC_DIR = the csrc directory to compile, default this tree's -- a git worktree of an experiment branch works too)
cd "${SRC_DIR:-/root/repo/revo_amd/csrc}" && mkdir -p /root/repo/profiles/build /tmp/t/v$SUFFIX
for f in revo_pyramid revo_track revo_host revo_vo revo_pipeline revo_comm; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $EXTRA_DEFS $( [ $f = revo_track ] && echo -mllvm -disable-machine-licm ) -w -c $f.hip -o /tmp/t/v$SUFFIX/$f.o -Rpass-analysis=kernel-resource-usage 2>/tmp/t/v$SUFFIX/$f.remarks & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/profiles/build/librevo_hip_var${SUFFIX}.so /tmp/t/v$SUFFIX/revo_pyramid.o /tmp/t/v$SUFFIX/revo_track.o /tmp/t/v$SUFFIX/revo_host.o /tmp/t/v$SUFFIX/revo_vo.o /tmp/t/v$SUFFIX/revo_pipeline.o /tmp/t/v$SUFFIX/revo_comm.o -ldl
cat /tmp/t/v$SUFFIX/*.remarks | grep -E 'Function Name|ScratchSize|VGPRs:' | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - | grep -E "${SHOW:-k_canny_nms|k_hyst|k_track}" | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-150
