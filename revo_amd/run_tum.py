"""`python -m revo_amd.run_tum <settings.yaml> <dataset.yaml>` -- the reference's command line
(main.cpp:22-47: REVO <settings.yaml> <dataset.yaml>) for TUM-layout datasets: runs the sequential
VO on one GPU and writes poses_<dataset>.txt in TUM format (system.cpp:48-49,76-80)."""
import os
import sys
import time

import numpy as np


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) < 2:
        print("usage: python -m revo_amd.run_tum <settings.yaml> <dataset.yaml> [device] [--save-model DIR] [--decoders N]")
        return 2
    from . import api, config, ply, synth, tum, vo
    model_dir = None
    if "--save-model" in argv:  # MapDrawer::saveModel (MapDrawer.h:97-170): outputPcl.ply + outputKf.ply
        i = argv.index("--save-model")
        model_dir = argv[i + 1]
        argv = argv[:i] + argv[i + 2:]
    decoders = None  # PNG decoder processes (tum.DecodePool); 0 = decode on the IO thread itself, like the reference
    if "--decoders" in argv:
        i = argv.index("--decoders")
        decoders = int(argv[i + 1])
        argv = argv[:i] + argv[i + 2:]
    from .settings import OptimizerSettings
    trk_settings, use_edge_filter, sysd = config.load_settings_yaml(argv[0])
    pyr_settings, io = config.load_dataset_yaml(argv[1])
    device = int(argv[2]) if len(argv) > 2 else 0
    trk_settings.optimizerSettings = OptimizerSettings(use_edge_filter=use_edge_filter)
    for ds in io["datasets"]:
        folder = os.path.join(io["main_folder"], ds)
        cam = api.CameraPyr(pyr_settings, device=device)
        drawer = ply.ModelExporter() if model_dir else None
        drv = vo.REVO(pyr_settings, trk_settings, cameraPyr=cam, depth_scale_factor=io["depth_scale_factor"],
                      mapDrawer=drawer, generate_dense_pcl=sysd["do_generate_dense_pcl"])
        nd = tum.default_decoders() if decoders is None else decoders
        rows = tum.read_associate(os.path.join(folder, io["associate"]), skip_first_n_frames=io["skip_first_n_frames"],
                                  read_n_images=io["read_n_images"])
        t0 = time.perf_counter()
        if nd >= 1:
            # iowrapperRGBD.cpp:301-333 on `nd` cores: the decoders fill a page-locked ring the IO thread submits from in place
            with tum.DecodePool(folder, rows, pyr_settings.width, pyr_settings.height, workers=nd,
                                use_depth_timestamp=bool(io["use_depth_timestamp"])) as pool:
                res = drv.run(pool)
        else:
            res = drv.run(tum.frames(folder, io["associate"], bool(io["use_depth_timestamp"]),
                                     skip_first_n_frames=io["skip_first_n_frames"], read_n_images=io["read_n_images"]))
        dt = time.perf_counter() - t0
        name = os.path.basename(os.path.normpath(ds)) or "dataset"
        if sysd["do_output_poses"]:
            with open("poses_%s.txt" % name, "w") as f:
                f.write("\n".join(drv.tum_lines()) + "\n")
        print("-----VO Report-----\nFrames Tracked: %d\nKeyframes Tracked: %d\nframes/s (incl. PNG decode, %s): %.1f"
              % (len(res), drv.nKeyFrames, ("%d decoder processes" % nd) if nd >= 1 else "decoded on the IO thread", len(res) / dt))
        if drawer is not None:
            out = drawer.saveModel(os.path.join(model_dir, name) if len(io["datasets"]) > 1 else model_dir)
            print("model: %d points of %d keyframes -> %s, %s" % (drawer.nPts, len(drawer.vpKfsF), out[0], out[1]))
        gt_file = os.path.join(folder, "groundtruth.txt")  # positions only: ATE (the RPE evaluator needs full poses)
        if os.path.exists(gt_file):
            gt = tum.read_groundtruth_positions(gt_file)
            est, ref = [], []
            for ts, M in drv.poses:
                if round(ts, 6) in gt:
                    est.append(M)
                    G = np.eye(4)
                    G[:3, 3] = gt[round(ts, 6)]
                    ref.append(G)
            if len(est) > 2:
                print("ATE RMSE vs groundtruth.txt: %.4f m over %d poses" % (synth.ate_rmse(est, ref), len(est)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
