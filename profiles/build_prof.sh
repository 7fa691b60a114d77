#!/bin/bash
# profiling build of the library (-DREVO_TRACK_PROFILE) -> profiles/build/librevo_hip_prof${SUFFIX}.so
cd /root/repo/revo_amd/csrc && mkdir -p /root/repo/profiles/build /tmp/t
for f in revo_pyramid revo_track revo_host revo_vo revo_pipeline revo_comm; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DREVO_TRACK_PROFILE $EXTRA_DEFS $( [ $f = revo_track ] && echo -mllvm -disable-machine-licm ) -w -c $f.hip -o /tmp/t/$f.prof.o & done; wait
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/profiles/build/librevo_hip_prof${SUFFIX}.so /tmp/t/revo_pyramid.prof.o /tmp/t/revo_track.prof.o /tmp/t/revo_host.prof.o /tmp/t/revo_vo.prof.o /tmp/t/revo_pipeline.prof.o /tmp/t/revo_comm.prof.o -ldl
