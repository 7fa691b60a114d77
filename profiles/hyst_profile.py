"""k_hyst phase cycles per level-0 frame (a -DREVO_HYST_PROFILE build prints one line per frame from the kernel):
REVO_HIP_SO=profiles/build/librevo_hip_var_hp.so python profiles/hyst_profile.py"""
import sys
import numpy as np
sys.path.insert(0, ".")


def main():
    import torch
    from revo_amd import api, synth
    from revo_amd.settings import ImgPyramidSettings, TrackerSettings
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    n = 32
    pairs = [synth.make_pair(i, s) for i in range(n)]
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).cuda()
    dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).cuda()
    bt = api.BatchTracker(cam, n)
    bt.build(bgr.data_ptr(), dep.data_ptr())
    bt.sync()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
