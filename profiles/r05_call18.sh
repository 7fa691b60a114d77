#!/bin/bash
# round 5, GPU call 18: k_canny_nms4 skips the four per-pixel comparisons of a row when no lane of the wavefront holds a pixel above
# the low threshold (wave-uniform branch, same bits): bit-exactness, then A/B against the kernel without the skip
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r5c18; mkdir -p $O
( time timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variants.py -m gpu -x -q ) > $O/pytest.log 2>&1; grep -E "passed|failed|error" $O/pytest.log | head -3
timeout 900 python profiles/ab_bench.py --runs 2 skip= noskip=profiles/build/librevo_hip_var_noskip.so 2>&1 | tee $O/ab_nms_skip.txt
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import torch
from revo_amd import api, synth
from revo_amd.settings import ImgPyramidSettings, TrackerSettings
def main():
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    pairs = [synth.make_pair(i, s) for i in range(8)]
    cam = api.CameraPyr(s); api.TrackerNew(TrackerSettings(), s, cam)
    bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")] * 4)).cuda()
    dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")] * 4)).cuda()
    bt = api.BatchTracker(cam, 32)
    print([(n, round(us, 1)) for n, us in bt.profile_build(bgr.data_ptr(), dep.data_ptr(), reps=5)])
if __name__ == "__main__":
    main()
PY
