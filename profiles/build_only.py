"""Six builds of the bench batch (64 frames, 640x480 x 4 levels) on one stream, nothing else running: for a rocprofv3
--kernel-trace --stats run that shows every build kernel's duration alone (profiles/r06_call9.sh)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def main():
    from revo_amd.settings import ImgPyramidSettings, TrackerSettings
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    n = 32
    rend = [bench.render_pair((sd, 640, 480, 4)) for sd in range(n)]  # (serial: profilers deadlock on a forked pool)
    import torch
    from revo_amd import api
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bgr = torch.from_numpy(np.stack([r[k] for r in rend for k in (0, 2)])).cuda()
    dep = torch.from_numpy(np.stack([r[k] for r in rend for k in (1, 3)])).cuda()
    bt = api.BatchTracker(cam, n)
    for _ in range(6):
        bt.build(bgr.data_ptr(), dep.data_ptr())
        bt.prepare()
        bt.sync()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
