"""TUM RGB-D dataset front-end (io/iowrapperRGBD.cpp:257-333) without OpenCV: 'associate.txt'
lines 'ts rgb_path ts depth_path', 8-bit RGB PNGs and 16-bit depth PNGs (metres = raw / 5000).
Decoding uses PIL; the u16 -> float conversion of iowrapperRGBD.cpp:326-327 is fused into the
device-side pyramid build (revo_pyramid_create_u16 / depth_scale_factor)."""
import os

import numpy as np


def read_associate(path, skip_first_n_frames=0, read_n_images=None):
    """-> list of (rgb_ts, rgb_file, depth_ts, depth_file); '#' comments and blank lines ignored."""
    out, n = [], 0
    for line in open(path):
        line = line.strip()
        if not line or line[0] == "#":
            continue
        n += 1
        if n <= skip_first_n_frames:
            continue
        p = line.split()
        if len(p) < 4:
            raise ValueError("bad associate line: %r" % line)
        out.append((float(p[0]), p[1], float(p[2]), p[3]))
        if read_n_images is not None and len(out) > read_n_images:  # `if (nFrames > READ_N_IMAGES) break;` after the
            break                                                      # push: N + 1 frames (iowrapperRGBD.cpp:291)
    return out


def load_frame(folder, rgb_file, depth_file):
    """-> (BGR8 [H,W,3], depth uint16 [H,W]) like cv::imread(rgb) / imread(depth, UNCHANGED)."""
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(folder, rgb_file)).convert("RGB"), np.uint8)
    dimg = Image.open(os.path.join(folder, depth_file))
    depth = np.asarray(dimg)
    if depth.dtype != np.uint16:
        depth = depth.astype(np.uint16)
    return np.ascontiguousarray(rgb[..., ::-1]), np.ascontiguousarray(depth)


def frames(folder, associate="associate.txt", use_depth_timestamp=False, **kw):
    """Generator of (bgr, depth_u16, timestamp)."""
    for rts, rf, dts, df in read_associate(os.path.join(folder, associate), **kw):
        bgr, depth = load_frame(folder, rf, df)
        yield bgr, depth, (dts if use_depth_timestamp else rts)


def write_synthetic_dataset(folder, seq, depth_scale=5000.0):
    """Writes a TUM-layout dataset (rgb/*.png, depth/*.png, associate.txt, groundtruth.txt) from
    revo_amd.synth.make_sequence frames: the stand-in for fr1/desk where no data is on disk."""
    from PIL import Image
    os.makedirs(os.path.join(folder, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(folder, "depth"), exist_ok=True)
    with open(os.path.join(folder, "associate.txt"), "w") as fa, open(os.path.join(folder, "groundtruth.txt"), "w") as fg:
        fa.write("# rgb depth (synthetic)\n")
        for bgr, depth, ts, T in seq:
            name = "%.6f.png" % ts
            Image.fromarray(np.ascontiguousarray(bgr[..., ::-1])).save(os.path.join(folder, "rgb", name))
            raw = np.clip(np.rint(depth.astype(np.float64) * depth_scale), 0, 65535).astype(np.uint16)
            Image.fromarray(raw).save(os.path.join(folder, "depth", name))
            fa.write("%.6f rgb/%s %.6f depth/%s\n" % (ts, name, ts, name))
            t = T[:3, 3]
            fg.write("%.6f %.9f %.9f %.9f 0 0 0 1\n" % (ts, t[0], t[1], t[2]))


def read_groundtruth_positions(path):
    out = {}
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        p = line.split()
        out[round(float(p[0]), 6)] = np.array([float(p[1]), float(p[2]), float(p[3])])
    return out
