"""Builds profiles/rNN_pmc_summary.json from two rocprofv3 --pmc rocpd databases (FETCH_SIZE pass, WRITE_SIZE pass):
HBM bytes per launch per kernel, FETCH_SIZE doubled (gfx950: 128-B requests tallied as 64 B,
MI355X_MICROARCH.md "HBM") and calibrated on k_gray_depth, whose bytes are known.
usage: pmc_summary.py <fetch.db> <write.db> <pairs> <width> <height> <command string> [borrow] > summary.json
(borrow: the build borrowed the f32 depth input, so k_gray_depth only reads BGR and writes gray)"""
import collections
import hashlib
import json
import os
import sqlite3
import sys


KERNEL_SOURCES = ("revo_pyramid.hip", "revo_track.hip", "revo_dev.h", "revo_div.h")


def kernel_sha16():
    """the same hash bench.py computes: sha256 over the kernel translation units and their headers (names + contents)"""
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "revo_amd", "csrc")
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    cols = [d[0] for d in db.execute("select * from counters_collection limit 1").description]
    kn = "kernel_name" if "kernel_name" in cols else "name"
    acc, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for k, c, v, d in db.execute("select %s, counter_name, value, dispatch_id from counters_collection" % kn):
        if c != counter:
            continue
        short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[short] += v
        cnt[short].add(d)
    return {k: acc[k] / len(cnt[k]) for k in acc}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
pairs, w, h = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
out = {"command": sys.argv[6], "kernel_sha16": kernel_sha16(), "commit": os.environ.get("REVO_COMMIT"), "unit_note": "FETCH_SIZE / WRITE_SIZE are KB per launch (rocprofv3); hbm_bytes_per_launch = "
       "(2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, checked on k_gray_depth"}
for k in sorted(set(fetch) | set(write)):
    f, wr = fetch.get(k, 0.0), write.get(k, 0.0)
    out[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": wr, "hbm_bytes_per_launch": (2 * f + wr) * 1024.0}
npix = w * h
rb, wb = (3, 1) if len(sys.argv) > 7 and sys.argv[7] == "borrow" else (7, 5)
out["calibration_k_gray_depth"] = {"known_read_KB": 2 * pairs * npix * rb / 1024.0, "known_write_KB": 2 * pairs * npix * wb / 1024.0,
                                   "FETCH_SIZE_KB": fetch.get("k_gray_depth"), "WRITE_SIZE_KB": write.get("k_gray_depth")}
print(json.dumps(out, indent=1))
