#!/bin/bash
python profiles/single_pair_phases.py 2>&1 | tail -1
cp build_variants/prof.so revo_amd/librevo_hip.so
python profiles/single_pair_phases.py 2>&1 | tail -1
for c in 4 2 1; do REVO_TRACK_CLUSTER=$c python profiles/single_pair_phases.py 2>&1 | tail -1; done
