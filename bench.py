#!/usr/bin/env python
"""bench.py -- headline benchmark of the REVO hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input that is
already resident in HBM: for each of the `pairs` independent frame-pairs of this
rank, build BOTH pyramids (gray, pyrDown, Canny, fill-in, 3-D edge list), promote
the reference frame to a keyframe (exact EDT + gradient table) and run the full
coarse-to-fine TrackerNew::trackFrames; with N > 1 ranks the step ends with one
RCCL all_gather of the 96-byte pair records.  Workload = BASELINE.json configs[2]
at N = 1 (synthetic 640x480, 4-level pyramid, 32 frame-pairs in flight) and
configs[4] at N = 8 (256 pairs sharded 32 per GPU): weak scaling.  configs[1]
(TUM fr1/desk, single stream) cannot run here: no TUM data, no network.

value = tracked frames (= frame-pairs) per second, whole job, max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


KERNEL_SOURCES = ("revo_pyramid.hip", "revo_track.hip", "revo_dev.h", "revo_div.h")  # everything the device code is compiled from


def kernel_sha16():
    """Identity of the DEVICE code a measurement was taken with: sha256 over the kernel translation units and the headers they
    include (names + contents).  profiles/pmc_summary.py stamps it into the PMC summary; a summary taken with other kernels is
    not quoted as `traffic` (host-side edits -- revo_host.hip, revo_pipeline.hip -- do not change a kernel's bytes per launch)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "revo_amd", "csrc")
    for name in KERNEL_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes
    expose 256 hardware threads but grant 16 CPUs: cpu.max = "1600000 100000")."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    return model, allowed


def resident_gate_state(device):
    """The library's resident gate (revo_host.hip): workgroups of tracker grids that have started / gates that gave up waiting /
    workgroups enqueued.  A healthy run has no time-outs, and census == enqueued once the device is idle."""
    import ctypes as C
    from revo_amd import _lib
    L = _lib.lib()
    if not hasattr(L, "revo_debug_census_"):
        return None
    o = (C.c_uint * 3)()
    L.revo_debug_census_.argtypes = [C.c_int, C.POINTER(C.c_uint)]
    if L.revo_debug_census_(int(device), o) != 0:
        return None
    return {"census": int(o[0]), "timeouts": int(o[1]), "enqueued": int(o[2])}


def baseline_config(w, h, levels, pairs, world):
    """Which BASELINE.json configuration a run IS, by geometry, not by world size alone."""
    if (w, h, levels) == (640, 480, 4):
        if world == 1 and pairs == 32:
            return "BASELINE configs[2]: synthetic 640x480 stream, 1 MI355X, batch = 32 frame-pairs"
        if world == 8 and pairs == 32:
            return "BASELINE configs[4]: batch = 256 frame-pairs sharded over 8 MI355X, RCCL gather"
        return "BASELINE configs[2]/[4] geometry with %d pairs per GPU on %d GPU(s)" % (pairs, world)
    if (w, h, levels) == (1280, 960, 5):
        return "BASELINE configs[3]: synthetic 1280x960, 5-level pyramid, 1 MI355X" + ("" if world == 1 else " (here on %d GPUs)" % world)
    return "not a BASELINE configuration: %dx%d, %d levels" % (w, h, levels)


def render_pair(args):
    seed, w, h, levels = args
    from revo_amd import synth
    from revo_amd.settings import ImgPyramidSettings
    s = ImgPyramidSettings.scaled(w, h, levels)
    p = synth.make_pair(seed, s)
    return p["ref"][0], p["ref"][1], p["curr"][0], p["curr"][1], p["T_ref_curr"]


def nearest_positions(gt, stamps, max_dt=0.02):
    """ground-truth positions at the estimate's time stamps (nearest within max_dt seconds, like TUM's associate.py)"""
    keys = np.array(sorted(gt))
    out = []
    for ts in stamps:
        j = int(np.searchsorted(keys, ts))
        best = min((k for k in (j - 1, j) if 0 <= k < len(keys)), key=lambda k: abs(keys[k] - ts), default=None)
        out.append(gt[keys[best]] if best is not None and abs(keys[best] - ts) <= max_dt else None)
    return out


def run_tum_stream(a, device):
    """The sequential stream on a TUM-layout folder: frames/s (PNG decode excluded: frames are decoded first, like the
    synthetic sweep), keyframes, ATE vs groundtruth.txt when present, and -- unless --cpu-baseline off -- the oracle on
    the first frames of the same input: trajectory difference GPU vs oracle (the 'ATE within 1 mm of the reference
    trajectory on identical inputs' check of BASELINE configs[0])."""
    from revo_amd import api, synth, tum, vo
    from revo_amd.settings import ImgPyramidSettings
    if not os.path.exists(os.path.join(a.tum_dir, "associate.txt")):
        return {"error": "no associate.txt in %s" % a.tum_dir}
    s3 = ImgPyramidSettings()  # config/dataset_tum1.yaml: fr1 intrinsics, 3 levels
    frames = list(tum.frames(a.tum_dir, read_n_images=(a.tum_frames - 1) if a.tum_frames > 0 else None))
    if len(frames) < 3 or frames[0][0].shape[:2] != (s3.height, s3.width):
        return {"error": "need >= 3 frames of %dx%d" % (s3.width, s3.height)}
    cam = api.CameraPyr(s3, device=device)
    vo.REVO(s3, cameraPyr=cam, depth_scale_factor=5000.0).run(frames[:6])
    runs = []
    for _ in range(3):
        drv = vo.REVO(s3, cameraPyr=cam, depth_scale_factor=5000.0)
        t0 = time.perf_counter()
        drv.run(frames)
        runs.append(time.perf_counter() - t0)
    out = {"folder": os.path.basename(os.path.normpath(a.tum_dir)), "frames": len(frames), "levels": 3,
           "frames_per_s": len(frames) / min(runs), "frames_per_s_runs": [len(frames) / t for t in runs],
           "keyframes": drv.nKeyFrames, "note": "decoded frames in host memory -> revo_vo_* (H2D + build on the IO thread)"}
    # ... and from the PNG files: tum.DecodePool (decoder processes -> a page-locked ring the IO thread submits from in place),
    # what `python -m revo_amd.run_tum` does; the single-process reader next to it (iowrapperRGBD.cpp:301-333 as the reference runs it)
    rows = tum.read_associate(os.path.join(a.tum_dir, "associate.txt"), read_n_images=(a.tum_frames - 1) if a.tum_frames > 0 else None)
    nd = tum.default_decoders()
    dec_runs = []
    for _ in range(2):
        drv2 = vo.REVO(s3, cameraPyr=cam, depth_scale_factor=5000.0)
        t0 = time.perf_counter()
        with tum.DecodePool(a.tum_dir, rows, s3.width, s3.height, workers=nd) as pool:
            pinned = pool.pinned
            pool.warm()
            t0 = time.perf_counter()  # (decoder start-up excluded: paid once per sequence)
            drv2.run(pool)
            dec_runs.append(time.perf_counter() - t0)
    same = len(drv2.poses) == len(drv.poses) and all(np.array_equal(p[1], q[1]) for p, q in zip(drv2.poses, drv.poses))
    t0 = time.perf_counter()
    n1 = min(len(rows), 60)
    for r in rows[:n1]:
        tum.load_frame(a.tum_dir, r[1], r[3])
    one_decoder = n1 / (time.perf_counter() - t0)
    out.update({"frames_per_s_incl_decode": len(rows) / min(dec_runs), "frames_per_s_incl_decode_runs": [len(rows) / t for t in dec_runs],
                "decoder_processes": nd, "decode_ring_page_locked": bool(pinned), "poses_identical_to_predecoded_run": bool(same),
                "one_decoder_frames_per_s": one_decoder})
    gt_file = os.path.join(a.tum_dir, "groundtruth.txt")
    if os.path.exists(gt_file):
        gt = tum.read_groundtruth_positions(gt_file)
        ref = nearest_positions(gt, [ts for ts, _ in drv.poses])
        est = [M for (ts, M), g in zip(drv.poses, ref) if g is not None]
        refm = []
        for g in ref:
            if g is not None:
                G = np.eye(4)
                G[:3, 3] = g
                refm.append(G)
        if len(est) > 2:
            out["ate_rmse_vs_groundtruth_m"] = synth.ate_rmse(est, refm)
            out["ate_poses"] = len(est)
    if a.cpu_baseline != "off":
        from oracle import ro
        n = min(len(frames), 40)
        ovo = ro.VO(s3)
        t0 = time.perf_counter()
        o_poses = [ovo.push(bgr, ro.u16_to_depth(raw, 5000.0), ts)[0] for bgr, raw, ts in frames[:n]]
        out["cpu_oracle_frames_per_s_1core"] = n / (time.perf_counter() - t0)
        out["trajectory_rmse_gpu_vs_oracle_m"] = synth.ate_rmse([M for _, M in drv.poses[:n]], o_poses)
        out["oracle_frames"] = n
    del drv
    return out


def dry_run_cpu(a, world, rank, json_fd):
    """The rank plumbing of this file without a GPU (gloo): spawn/rendezvous (done by the caller), static shard, one
    96-byte record per pair from a stubbed step, the gather schedule (--gather-every) with the STREAM semantics of the real
    loop -- the collective of a window is issued asynchronously behind the window's last step and is waited for only when
    its tracker stream is used again, two steps later --, max-over-ranks timing, ONE JSON line from rank 0.
    No measurement: value is null and the line says dry_run.  --dry-run-step-ms makes the stubbed step take time and
    --dry-run-delay-rank/-ms make one rank late in one step: `ms_loop_per_rank` (every rank's clock around its own step
    loop, before the final drain) then shows whether the others were stalled behind it."""
    import torch
    import torch.distributed as dist
    from revo_amd import parallel
    if world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(parallel.free_port()))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    every = a.gather_every or parallel.gather_every_default(world)
    seeds = parallel.shard_pairs(world * a.pairs, rank, world)
    n_slots = a.warmup + a.steps
    nrec = parallel.RECORD_BYTES // 4
    rec_all = np.zeros((n_slots, a.pairs, nrec), np.float32)  # one record slot per step, like the real loop
    local_all = torch.from_numpy(rec_all.view(np.uint8).reshape(n_slots, -1))
    outs = {}
    pending = []  # (stream index, work handle, out tensor, first step, n steps)
    checked = [0]
    # A LOGICAL clock next to the wall clock (the sleep-based check of "a late rank does not serialise the others"
    # depends on the host's scheduling): a step costs --dry-run-step-ms of logical time (+ the delay on the late rank), every
    # record carries the logical time at which its step ended, a collective completes at the maximum of the ranks' logical
    # issue times, and waiting for it moves the waiting rank's clock there.  Deterministic: the schedule property is exact.
    lt = [0.0]
    min_issue_to_wait = [None]  # fewest steps between issuing a collective and first waiting for it (the after-grid slot: 2)

    def check(out, first, n):
        g = out.numpy().view(np.float32).reshape(world, n, a.pairs, nrec)
        for r in range(world):
            for k in range(n):
                if not (np.array_equal(g[r, k, :, 0], np.arange(r * a.pairs, (r + 1) * a.pairs, dtype=np.float32))
                        and np.all(g[r, k, :, 9] == r) and np.all(g[r, k, :, 10] == first + k)):
                    raise SystemExit("bench --dry-run-cpu: the gather did not return every rank's records of steps %d..%d in rank order"
                                     % (first, first + n - 1))
        checked[0] += n

    def wait_stream(si, t_now=None):
        for item in [x for x in pending if x[0] == si or si < 0]:
            item[1].wait()
            check(item[2], item[3], item[4])
            g = item[2].numpy().view(np.float32).reshape(world, item[4], a.pairs, nrec)
            lt[0] = max(lt[0], float(g[:, item[4] - 1, 0, 11].max()))  # every rank had issued it by then (logical time)
            if t_now is not None:
                gap = t_now - (item[3] + item[4] - 1)
                min_issue_to_wait[0] = gap if min_issue_to_wait[0] is None else min(min_issue_to_wait[0], gap)
            pending.remove(item)

    def issue(first, n, si):
        src = local_all[first:first + n].reshape(-1)
        out = outs.setdefault((n, si), torch.empty(world * src.numel(), dtype=torch.uint8))
        pending.append((si, dist.all_gather_into_tensor(out, src, async_op=True), out, first, n))

    def step(t, start):
        si = t % 2                       # the tracker stream of step t
        wait_stream(si, t)               # its grid sits behind whatever was enqueued on that stream before
        if a.dry_run_step_ms > 0:
            time.sleep(a.dry_run_step_ms * 1e-3)
            lt[0] += a.dry_run_step_ms
        if rank == a.dry_run_delay_rank and t == a.warmup + 2 and a.dry_run_delay_ms > 0:
            time.sleep(a.dry_run_delay_ms * 1e-3)
            lt[0] += a.dry_run_delay_ms
        rec_all[t, :, 0] = seeds         # what a tracker would write: the pair's global index ...
        rec_all[t, :, 9] = rank          # ... the rank that "tracked" it ...
        rec_all[t, :, 10] = t            # ... and the step
        rec_all[t, :, 11] = lt[0]        # ... and the logical time at which the step ended (the collective is issued then)
        win = parallel.gather_window(t, every, start)
        if win:
            issue(win[0], win[1], si)

    def phase(start, end):
        for t in range(start, end):
            step(t, start)
        loop = time.perf_counter()
        phase_lt[0] = lt[0]
        tail = parallel.gather_tail(end, every, start)  # what the last window did not cover travels in one final collective
        if tail:
            issue(tail[0], tail[1], (end - 1) % 2)
        wait_stream(-1)
        return loop

    phase_lt = [0.0]
    phase(0, a.warmup)
    dist.barrier()
    lt[0] = 0.0  # (float32 in the records: logical times stay small integers of milliseconds)
    min_issue_to_wait[0] = None
    t0 = time.perf_counter()
    loop_s = phase(a.warmup, n_slots) - t0  # this rank's own loop: nobody has been waited for beyond the stream order
    mine_s = time.perf_counter() - t0
    dist.barrier()
    per_rank = parallel.gather_floats(mine_s, world)
    loop_per_rank = parallel.gather_floats(loop_s, world)
    logical_per_rank = parallel.gather_floats(phase_lt[0], world)
    gap_per_rank = parallel.gather_floats(float(min_issue_to_wait[0] if min_issue_to_wait[0] is not None else -1), world)
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, world)
    seen, wsz = parallel.ranks_seen(world)
    if len(set(seen)) != world or wsz != world:
        raise SystemExit("bench --dry-run-cpu: the process group saw ranks %s (size %d) but WORLD_SIZE is %d" % (seen, wsz, world))
    if rank == 0:
        line = {"metric": "tracked frames/sec at 640x480, 4-level pyramid; ATE vs reference", "value": None, "unit": "frames/s",
                "dry_run": True, "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / max(1, a.steps) * 1e3,
                "ms_per_step_per_rank": [x / max(1, a.steps) * 1e3 for x in per_rank],
                "ms_loop_per_rank": [x * 1e3 for x in loop_per_rank],
                "logical_ms_loop_per_rank": logical_per_rank,  # the same loop on the logical clock: deterministic
                "min_steps_between_issue_and_first_wait_per_rank": [int(x) for x in gap_per_rank],
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "none (stubbed step)",
                "config": {"workload": "orchestration dry run on CPU (gloo): no kernels", "pairs_per_gpu": a.pairs,
                           "global_pairs": world * a.pairs},
                "collective": {"backend": "gloo", "executed_every_step": every == 1, "steps_per_collective": every,
                               "steps_gathered_and_checked": checked[0],
                               "bytes_per_rank": a.pairs * parallel.RECORD_BYTES * every,
                               "ranks_seen": seen, "world_size": wsz}}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--pairs", type=int, default=32, help="frame-pairs per GPU per step")
    ap.add_argument("--buffers", type=int, default=4,
                    help="batches in rotation: 4 = build(k+3) | edge lists + EDT(k+2) | tracker grids of steps k+1 and k (the library's "
                         "resident gate keeps two tracker grids in flight); 3 = round 3's shape; 2 = round 2's double buffering")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--input-cache", default="", help="npz path: load the rendered inputs from it if present, "
                    "else render and save (profilers deadlock on the forked render pool)")
    ap.add_argument("--render-procs", type=int, default=0, help="0 = auto")
    ap.add_argument("--no-overlap", action="store_true", help="single batch, single stream (no build/track overlap)")
    ap.add_argument("--single-stream-frames", type=int, default=60, help="0 = skip the sequential-VO side measurement")
    ap.add_argument("--tum-dir", default=os.environ.get("REVO_TUM_DIR", ""),
                    help="a TUM-layout sequence folder (rgb/, depth/, associate.txt[, groundtruth.txt]), e.g. "
                         "rgbd_dataset_freiburg1_desk: BASELINE configs[0]/[1] -- the sequential stream is then ALSO run "
                         "on it (shipped 3-level settings, raw u16 depth / 5000) and reported as `tum_stream`")
    ap.add_argument("--tum-frames", type=int, default=200, help="frames of --tum-dir to run (0 = all)")
    ap.add_argument("--skip-host-buffers", action="store_true", help="skip the host-buffer (H2D-inclusive) side measurement")
    ap.add_argument("--no-collective", action="store_true", help="N = 1 only: do not create the world-size-1 RCCL group")
    ap.add_argument("--input-batches", type=int, default=3,
                    help="distinct synthetic input batches rotated through the timed loop (3 x 138 MB at the default size: more "
                         "than the 256 MB Infinity Cache, so 'resident in HBM' cannot mean 'resident in the last-level cache')")
    ap.add_argument("--single-stream-runs", type=int, default=5, help="full-length runs of the sequential stream (median reported)")
    ap.add_argument("--coll", default="torch", choices=["torch", "native"],
                    help="who runs the path's only collective.  torch (default): torch.distributed's RCCL group, enqueued by this file in "
                         "the pipeline's after-grid slot.  native: the library's own communicator (revo_comm_*, RCCL loaded by "
                         "librevo_hip.so at run time) attached to the pipeline handle, which then enqueues the all-gather itself "
                         "(revo_pipeline_set_comm: what a C++ host uses); torch.distributed only carries the control plane then "
                         "(gloo: the 128-byte id, barriers, the max over ranks)")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="steps per RCCL all_gather (the records of that many steps travel in one collective, in the after-grid slot "
                         "of the last of them); 0 = the default, 2 at every N (ranks then meet every second step only: a rank may lag "
                         "a full step without stalling the others' tracker streams; N = 1 runs the same schedule); 1 = one per step")
    ap.add_argument("--dry-run-delay-rank", type=int, default=-1, help="--dry-run-cpu: this rank sleeps --dry-run-delay-ms in one step")
    ap.add_argument("--dry-run-delay-ms", type=float, default=0.0)
    ap.add_argument("--dry-run-step-ms", type=float, default=0.0, help="--dry-run-cpu: duration of the stubbed step")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="orchestration check without a GPU: the rank plumbing of this file (spawn, rendezvous, shard, records, "
                         "gather, max-over-ranks, one JSON line from rank 0) on the gloo backend with a stubbed step; the line "
                         "carries dry_run=true and no measurement")
    a = ap.parse_args()

    from revo_amd import parallel
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU,
        # env:// rendezvous on 127.0.0.1); rank 0 prints the JSON line on this process's stdout
        codes = parallel.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], a.gpus)
        bad = [(r, c) for r, c in enumerate(codes) if c != 0]
        if bad:
            raise SystemExit("bench: ranks failed (rank, exit code): %s" % bad)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries exactly ONE line, the JSON record: native libraries print there too (RCCL's version
    # banner is flushed to fd 1 at exit), so fd 1 points at stderr until the record is written to the real one
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if "WORLD_SIZE" in os.environ and a.gpus != world and rank == 0:
        print("bench: --gpus %d ignored, the launcher set WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr)
    if a.dry_run_cpu:
        return dry_run_cpu(a, world, rank, json_fd)

    # ---- synthetic input (rendered on the host BEFORE any GPU state exists) ----
    from revo_amd.settings import ImgPyramidSettings, TrackerSettings
    hist = tuple([20, 10, 5] + [0] * 3) if a.width == 640 else tuple([20, 10, 5, 0, 0, 0])
    s = ImgPyramidSettings.scaled(a.width, a.height, a.levels, hist_patch=hist)
    seeds = parallel.shard_pairs(world * a.pairs, rank, world)  # static block partition: rank g owns [g*B/G, (g+1)*B/G)
    # input batch j of this rank: the same shard of a disjoint seed range (batch 0 = seeds 0 .. world*pairs-1)
    nin = max(1, a.input_batches)
    jobs = [(j * world * a.pairs + sd, a.width, a.height, a.levels) for j in range(nin) for sd in seeds]
    nproc = a.render_procs or max(1, min(16, usable_cpus() // max(1, world), len(jobs)))
    t0 = time.time()
    cache = a.input_cache and ("%s.r%d.npz" % (a.input_cache, rank))
    if cache and os.path.exists(cache):
        z = np.load(cache)
        if len(z["rb"]) < len(jobs):
            raise SystemExit("bench: %s holds %d pairs, %d are needed (delete it or lower --input-batches)" % (cache, len(z["rb"]), len(jobs)))
        if tuple(z["rb"].shape[1:]) != (a.height, a.width, 3) or tuple(z["rd"].shape[1:]) != (a.height, a.width):
            raise SystemExit("bench: %s holds %dx%d frames, this run is %dx%d (one cache file per geometry)"
                             % (cache, z["rb"].shape[2], z["rb"].shape[1], a.width, a.height))
        rendered = [(z["rb"][i], z["rd"][i], z["cb"][i], z["cd"][i], z["gt"][i]) for i in range(len(jobs))]
    elif nproc > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(nproc) as pool:
            rendered = pool.map(render_pair, jobs)
    else:
        rendered = [render_pair(j) for j in jobs]
    t_render = time.time() - t0
    if cache and not os.path.exists(cache):
        np.savez(cache, rb=np.stack([r[0] for r in rendered]), rd=np.stack([r[1] for r in rendered]),
                 cb=np.stack([r[2] for r in rendered]), cd=np.stack([r[3] for r in rendered]),
                 gt=np.stack([r[4] for r in rendered]))
    rendered_all = rendered
    bgrs = [np.stack([r[k] for r in rendered_all[j * a.pairs:(j + 1) * a.pairs] for k in (0, 2)]) for j in range(nin)]
    deps = [np.stack([r[k] for r in rendered_all[j * a.pairs:(j + 1) * a.pairs] for k in (1, 3)]) for j in range(nin)]
    gts = [[r[4] for r in rendered_all[j * a.pairs:(j + 1) * a.pairs]] for j in range(nin)]
    rendered = rendered_all[:a.pairs]  # input batch 0: what the CPU baseline and the host-buffer legs use
    bgr, dep = bgrs[0], deps[0]

    import torch
    import torch.distributed as dist
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench: rank %d needs GPU %d but only %d are visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from revo_amd import api, synth
    cam = api.CameraPyr(s, device=local_rank)
    api.TrackerNew(TrackerSettings(), s, cam)

    # ---- side measurement, taken FIRST: ONE sequential stream through the host-buffer API (BASELINE configs[1]
    # stand-in: no TUM data here).  Frame N depends on frame N-1, so this is latency-bound and PCIe-inclusive; it is
    # reported next to, never as, `value`.  It runs before the process group exists because a sequential VO process
    # has no collective, and an initialised RCCL communicator slows the latency-bound stream down for the rest of
    # the process's life, destroyed or not (measured on the same box: 3.5 k frames/s before init, 2.35 k after).
    seq_gpu = None
    if rank == 0 and world == 1 and a.single_stream_frames > 0:
        from revo_amd import vo
        n_seq = a.single_stream_frames
        # a seeded synthetic camera sweep (TUM-like inter-frame motion: ~4 mm, 1 deg per frame)
        seq = synth.make_sequence(7, s, n_seq, max_t=0.01, max_rot_deg=0.4, bias=[0.004, 0, 0, 0, np.deg2rad(1.0), 0])
        # the frames sit in PAGE-LOCKED host memory, as a decoder thread that owns its buffers leaves them: revo_vo_submit then
        # lets the DMA engine read them in place (no host-side staging copy; pageable frames still work, through a staging copy)
        def pin(x):
            return torch.from_numpy(np.ascontiguousarray(x)).pin_memory().numpy()
        stream_frames = [(pin(f[0]), pin(f[1]), f[2]) for f in seq]
        # warm-up: two full-length runs (pools, first touch of every slot, code paths).  One run in 15-25 is 20-50 % slow: once in some ten
        # thousand frames a HIP call of the IO thread (hipMemcpyAsync, the in-place upload of a page-locked frame, in five observed cases
        # out of six; a build's kernel launches once) does not return for 6-13 ms, the IO thread sits in it,
        # the four queued pyramids run out and the consumer waits -- found in round 6 with per-pose / per-submit time stamps and
        # per-section maxima inside the library (profiles/r06_slow_run_probe.txt; REVO_H2D_KERNEL=1 uploads with a copy kernel instead:
        # no such run in 900, but a 6-14 % lower median).  7 ms in ~18 000 frames is 0.2 % of a long stream; a 15-ms run shows it as
        # a slow run.  All runs are reported, the median is the figure.
        for _ in range(2):
            vo.REVO(s, cameraPyr=cam).run(stream_frames)
        runs = []
        import gc
        import ctypes
        from revo_amd import _lib as _revo_lib
        sec = (ctypes.c_ulonglong * 12)()
        _revo_lib.lib().revo_debug_section_max_(sec, 1)  # (reset: the maxima below belong to the timed runs)
        for _ in range(max(1, a.single_stream_runs)):  # all runs reported, the MEDIAN is the figure
            drv = vo.REVO(s, cameraPyr=cam)
            # (r03 / r04: the SECOND of the five runs was reproducibly ~30 % slow in every invocation -- a deterministic point in
            # the interpreter's allocation count, i.e. a generational collection inside the 16 ms run: collect before, not during)
            gc.collect()
            gc.disable()
            try:
                t0 = time.perf_counter()
                drv.run(stream_frames)  # IO thread builds pyramids into the queue, this thread tracks (system.cpp:96)
                runs.append(time.perf_counter() - t0)
            finally:
                gc.enable()
        rpe = synth.rpe_rmse([p[1] for p in drv.poses], [f[3] for f in seq])
        # the longest any section of a frame submission took inside the timed runs (the library keeps the maxima): a slow run shows
        # up here as ONE hipMemcpyAsync of several milliseconds (profiles/r06_slow_run_probe.txt)
        _revo_lib.lib().revo_debug_section_max_(sec, 0)
        sec_names = ["lock_and_pool", "pointer_attributes", "build_stream_waits_for_copies", "wait_for_copies", "enqueue_build",
                     "wait_for_queue_room", "copy_stream_waits", "hipMemcpyAsync_colour", "hipMemcpyAsync_depth", "hipEventRecord"]
        io_max_ms = {nm: sec[i] / 1e6 for i, nm in enumerate(sec_names)}
        seq_gpu = {"runs": runs, "io_max_ms": io_max_ms, "keyframes": drv.nKeyFrames, "rpe": rpe, "poses": [np.array(p[1]) for p in drv.poses],
                   "gt": [f[3] for f in seq],
                   "ate": synth.ate_rmse([p[1] for p in drv.poses], [f[3] for f in seq])}
        del drv

    # ---- BASELINE configs[0]/[1]: a TUM sequence, when one is on disk (none ships with this repo).  Same driver, the
    # shipped settings (640x480, 3 levels, config/dataset_tum1.yaml), raw 16-bit depth converted inside the build.
    tum_out = None
    if rank == 0 and world == 1 and a.tum_dir:
        tum_out = run_tum_stream(a, local_rank)

    # The path's only collective runs through RCCL at every N, N = 1 included (the world-size-1
    # path is the single-GPU CI of the multi-GPU job).
    group_error = None
    native = a.coll == "native"
    if native and a.no_collective:
        raise SystemExit("bench: --coll native and --no-collective contradict each other")
    if world > 1 or not a.no_collective:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(parallel.free_port()))
        try:
            if native:  # control plane only: the data-path collective is the library's
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        except Exception as e:  # noqa: BLE001 -- N = 1 can still be measured without the group; N > 1 cannot
            if world > 1 or native:
                raise
            group_error = "%s: %s" % (type(e).__name__, e)
    use_group = dist.is_initialized()
    ctl_dev = "cpu" if native else dev  # where the control-plane tensors (times, rank ids) live
    comm = None
    if native:
        uid = [api.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = api.Comm(cam, uid[0], world, rank)  # ncclCommInitRank: every rank

    # The pipelined step: `nbuf` batches in rotation over four streams -- build(t+3) | edge lists + keyframe EDT(t+2) | the
    # tracker grids of steps t+1 and t on two alternating streams (the library's resident gate keeps two grids in flight) --, all
    # of it owned by the library's pipeline handle (revo_pipeline_*).  (Round 4's bench-owned loop over the batch entry points,
    # `--shape bench`, was retired in round 6: its A/B role ended with profiles/r05_ab_lib_vs_bench.txt.)
    nbuf = 1 if a.no_overlap else max(2, a.buffers)
    d_bgrs = [torch.from_numpy(b).to(dev) for b in bgrs]
    d_deps = [torch.from_numpy(d).to(dev) for d in deps]
    d_bgr, d_dep = d_bgrs[0], d_deps[0]
    # every step keeps its own result records, so that ALL of them are checked after the timed region
    n_slots = a.warmup + a.steps
    d_res_all = torch.zeros(max(1, n_slots) * a.pairs * 96, dtype=torch.uint8, device=dev)
    d_ress = [d_res_all[i * a.pairs * 96:(i + 1) * a.pairs * 96] for i in range(max(1, n_slots))]
    d_res = d_ress[0]
    d_res_side = torch.zeros(a.pairs * 96, dtype=torch.uint8, device=dev)  # side measurements write here
    every = max(1, a.gather_every or parallel.gather_every_default(world))  # steps per collective
    # the timing events cost ~1.5 % of the step when every launch carries a pair (two marker packets in front of / behind
    # each tracker on its stream): every 4th launch of the timed region is timed (every launch when there are few)
    TIME_EVERY = int(os.environ.get('REVO_BENCH_TIME_EVERY', '4' if a.steps >= 16 else '1'))
    gathered = [None, None]  # the last collective: (tensor with every rank's records in rank order, (first step, n steps))
    d_alls = {}              # (steps in the window, stream) -> the collective's output buffer
    ev_grid = [torch.cuda.Event() for _ in range(max(8, every + 2))]  # grid of step t done (cross-stream order of a multi-step window)
    ext_streams = {}

    def ext(handle):
        if handle not in ext_streams:
            ext_streams[handle] = torch.cuda.ExternalStream(handle, device=dev)
        return ext_streams[handle]

    def after_grid(t, s_tr, start):
        """The path's only collective: 96 B x pairs (x steps of the window) per rank, RCCL (over xGMI at N > 1), enqueued on the
        tracker's stream right behind the grid of the window's last step -- the after-grid slot of the pipeline (a fifth active
        stream would end up behind another stream's kernels in one of HIP's hardware queues: DESIGN 3.0).  Inside the timed
        region (synchronize + barrier below)."""
        if not use_group or native:  # (native: the pipeline handle enqueues the window's all-gather itself)
            return
        ev_grid[t % len(ev_grid)].record(s_tr)
        win = parallel.gather_window(t, every, start)
        if win:
            issue_gather(win, s_tr)

    def issue_gather(win, s_tr):
        first, n = win
        for u in range(first, first + n - 1):  # the window's earlier grids ran on the other tracker stream
            s_tr.wait_event(ev_grid[u % len(ev_grid)])
        src = d_res_all[first * a.pairs * 96:(first + n) * a.pairs * 96]
        key = (n, s_tr.cuda_stream)
        if key not in d_alls:
            d_alls[key] = torch.zeros(world * n * a.pairs * parallel.RECORD_BYTES, dtype=torch.uint8, device=dev)
        with torch.cuda.stream(s_tr):
            gathered[0] = parallel.gather_records(src, world, out=d_alls[key])
        gathered[1] = win

    timing = [False]
    pipe = api.Pipeline(cam, a.pairs, depth=nbuf)
    pipe_info = pipe.info()
    NATIVE_RING = 4
    d_gathered = None
    if native:  # [ring][rank][step in window][pair] records, written by the handle's own all-gather
        d_gathered = torch.zeros(NATIVE_RING * world * every * a.pairs * parallel.RECORD_BYTES, dtype=torch.uint8, device=dev)
        pipe.set_comm(comm, every, d_gathered.data_ptr(), NATIVE_RING)
    native_last = [None]  # (slot, steps valid, first step) of the last window gathered
    counter = [0]
    phase_start = [0]

    def step():
        t = counter[0]
        counter[0] += 1
        ticket, sh = pipe.submit(d_bgrs[t % nin].data_ptr(), d_deps[t % nin].data_ptr(), d_ress[t % len(d_ress)].data_ptr())
        after_grid(t, ext(sh), phase_start[0])
    s_track = ext(pipe_info["streams"][0])
    s_tracks = [ext(h) for h in sorted(set(pipe_info["streams"][:2]), key=pipe_info["streams"].index)]
    s_build = ext(pipe_info["streams"][2])
    s_edts = [ext(pipe_info["streams"][3])] if pipe_info["streams"][3] != pipe_info["streams"][2] else []
    torch.cuda.set_stream(s_track)
    stream = s_track.cuda_stream
    assert stream != 0 and s_build.cuda_stream != 0

    def close_phase(end):
        """what the last window of a phase did not cover travels in one final collective"""
        tail = parallel.gather_tail(end, every, phase_start[0])
        if native:  # an incomplete last window travels in one final collective of the handle's (every rank calls it)
            valid, slot = pipe.flush_comm()
            if valid:
                native_last[0] = (slot, valid, end - valid)
            elif end - phase_start[0] >= every:
                nwin = pipe.info()["steps_submitted"] // every
                native_last[0] = ((nwin - 1) % NATIVE_RING, every, end - every)
            return
        if use_group and tail:
            issue_gather(tail, s_tracks[(end - 1) % len(s_tracks)])

    for _ in range(a.warmup):
        step()
    close_phase(a.warmup)
    torch.cuda.synchronize()
    if use_group:
        dist.barrier()
    torch.cuda.synchronize()
    timing[0] = True
    phase_start[0] = a.warmup
    pipe.time_tracker(TIME_EVERY)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    close_phase(n_slots)
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0  # this rank's own clock: its streams have drained, nobody has been waited for at a barrier yet
    if use_group:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timing[0] = False
    ms_per_rank = [x / a.steps * 1e3 for x in parallel.gather_floats(mine_s, world, device=ctl_dev)] if use_group else [mine_s / a.steps * 1e3]
    elapsed = parallel.max_over_ranks(elapsed, world, device=ctl_dev)
    d_res = d_ress[(counter[0] - 1) % len(d_ress)]
    if native:  # the window slot of the handle's last collective holds this rank's records of those steps at its rank offset
        if native_last[0] is None:
            raise SystemExit("bench: no native collective ran in the timed region")
        slot, nst, first = native_last[0]
        wb = every * a.pairs * parallel.RECORD_BYTES
        mine = d_gathered[(slot * world + rank) * wb:(slot * world + rank) * wb + nst * a.pairs * parallel.RECORD_BYTES]
        if not torch.equal(mine, d_res_all[first * a.pairs * 96:(first + nst) * a.pairs * 96]):
            raise SystemExit("bench: the native RCCL gather did not return this rank's records")
    elif use_group:  # the gathered buffer of the last collective holds this rank's records of its window at its rank offset
        got, win = gathered
        if got is None or got.numel() != world * win[1] * a.pairs * parallel.RECORD_BYTES:
            raise SystemExit("bench: gathered record buffer has the wrong size")
        wb = win[1] * a.pairs * 96
        mine = got[rank * wb:(rank + 1) * wb]
        if not torch.equal(mine, d_res_all[win[0] * a.pairs * 96:(win[0] + win[1]) * a.pairs * 96]):
            raise SystemExit("bench: the RCCL gather did not return this rank's records")
    pipe.drain()
    ms_live, n_live = pipe.tracker_ms()
    pipe.time_tracker(0)
    pipe_info = pipe.info()  # (steps_submitted = warm-up + timed steps)
    gate = resident_gate_state(local_rank)
    if gate and gate["timeouts"]:
        print("bench: WARNING: %d resident gates gave up waiting (census %d of %d enqueued workgroups): tracker grids were "
              "held back for the gate's time-out -- the line below does not describe a healthy run" % (gate["timeouts"], gate["census"], gate["enqueued"]), file=sys.stderr)
    # one batch for the measurements outside the timed region (stage split, k_track alone, point counts)
    bt = api.BatchTracker(cam, a.pairs)

    # ---- correctness of what was timed (never skipped work): the flags of EVERY step, poses vs ground truth
    all_res = api.results_from_buffer(d_res_all.cpu().numpy().tobytes(), n_slots * a.pairs)
    bad_flags = [(i // a.pairs, i % a.pairs, r["flags"]) for i, r in enumerate(all_res) if r["flags"] & (2 | 4 | 8)]
    if bad_flags:
        raise SystemExit("bench: the tracker flagged invalid results (step, pair, flags): %s" % bad_flags[:8])
    res = all_res[(n_slots - 1) * a.pairs:]
    # step t read input batch t mod nin (on batch object t mod nbuf): equal inputs must give equal bits, whichever step
    for t in range(nin, n_slots):
        x, y = all_res[(t - nin) * a.pairs:(t - nin + 1) * a.pairs], all_res[t * a.pairs:(t + 1) * a.pairs]
        if any(not (np.array_equal(p["R"], q["R"]) and np.array_equal(p["T"], q["T"])) for p, q in zip(x, y)):
            raise SystemExit("bench: the same inputs gave different poses in steps %d and %d" % (t - nin, t))
    errs = [synth.pose_error(r["R"], r["T"], g) for t in range(max(0, n_slots - nin), n_slots)
            for r, g in zip(all_res[t * a.pairs:(t + 1) * a.pairs], gts[t % nin])]
    rot_med = float(np.median([e[0] for e in errs]))
    tr_med = float(np.median([e[1] for e in errs]))

    # ---- roofline of the dominant kernel (k_track): HIP events on its stream around every launch of the
    # timed region (next to the other stream's build kernels); `kernel_ms_alone` re-times it with nothing else
    # running (revo_batch_time_tracker)
    # the pipeline handle's own event pairs (revo_pipeline_time_tracker: recorded inside the tracker chain, directly around the
    # grid), harvested after the drain
    ms_track = float(ms_live) if n_live else None
    n_timed = n_live
    # algorithmic bytes per launch, SURVEY 8(d): B_trk = sum_l E_l*N_l*(16 + 4*16) + init check 2*N_c*(16+4) -- the mean
    # over the input batches of the rotation (every one is rebuilt and tracked once more here, outside the timed
    # region, to read its point counts; `kernel_ms_alone` averages over the same inputs)
    b_trks, alone, npts_all, evals_all = [], [], [], []
    for j in range(nin):
        bt.build(d_bgrs[j].data_ptr(), d_deps[j].data_ptr(), stream=stream, borrow_depth=True)
        alone.append(bt.time_tracker(d_res.data_ptr(), reps=max(2, 10 // nin), stream=stream))
        torch.cuda.synchronize()
        rj = api.results_from_buffer(d_res.cpu().numpy().tobytes(), a.pairs)
        npts = np.zeros((a.pairs, a.levels), np.int64)
        for i in range(a.pairs):
            view = bt.frame(2 * i + 1, s)
            for lvl in range(a.levels):
                npts[i, lvl] = view.return3DEdges(lvl).shape[0]
        evals = np.array([r["evals"][: a.levels] for r in rj], np.int64)
        b_trks.append(float((evals * npts * 80).sum() + (2 * npts[:, a.levels - 1] * 20).sum()))
        npts_all.append(npts)
        evals_all.append(np.array([r["evals"] for r in rj], np.float64))
    b_trk = float(np.mean(b_trks))
    ms_track_alone = float(np.mean(alone))
    if ms_track is None:
        ms_track = ms_track_alone
    npts = np.concatenate(npts_all)
    evals = np.concatenate(evals_all)[:, : a.levels]
    achieved = b_trk / (ms_track * 1e-3) / 1e9  # GB/s

    # stage split (events around build-only / track-only, same stream)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    reps = 5
    torch.cuda.synchronize()
    tb = tk = 0.0
    for _ in range(reps):
        e0.record()
        bt.build(d_bgr.data_ptr(), d_dep.data_ptr(), stream=stream, borrow_depth=True)
        bt.prepare(stream=stream)  # the keyframes' EDT belongs to the build stage wherever it is run
        e1.record()
        bt.track_only(d_res.data_ptr(), stream=stream)
        e2.record()
        torch.cuda.synchronize()
        tb += e0.elapsed_time(e1)
        tk += e1.elapsed_time(e2)
    ms_build, ms_trk_stage = tb / reps, tk / reps

    # ---- what "32 frame-pairs in flight" means, stated three ways: `value` keeps `nbuf` batches in rotation
    # (declared in config.pipelining / pairs_resident); next to it the SAME step with ONE batch and nothing overlapped
    # (build -> keyframes -> tracker on one stream) and with TWO batches (the build of step k+1 next to the tracker of step
    # k, one tracker stream).  Same kernels, same inputs, no collective; 3 warm-up + 20 timed steps each, this rank only.
    def side_rate(nb):
        sp = api.Pipeline(cam, a.pairs, depth=nb)
        cnt = [0]

        def st():
            sp.submit(d_bgrs[cnt[0] % nin].data_ptr(), d_deps[cnt[0] % nin].data_ptr(), d_res_side.data_ptr())
            cnt[0] += 1
        for _ in range(3):
            st()
        torch.cuda.synchronize()
        t0s = time.perf_counter()
        for _ in range(20):
            st()
        torch.cuda.synchronize()
        rate = a.pairs * 20 / (time.perf_counter() - t0s)
        sp.close()
        return rate
    value_single = side_rate(1) if not a.no_overlap else a.pairs * a.steps / elapsed
    value_two = side_rate(2) if not a.no_overlap else None

    # HBM traffic of k_track per launch from the PMC passes recorded under profiles/ (same workload; counters cannot be
    # collected inside this process).  The summary carries the hash of the kernel sources it was taken with: a summary of other
    # sources is NOT quoted (the constant must not go stale silently).
    traffic, traffic_src, traffic_commit = None, None, None
    src_now = kernel_sha16()
    for pmc_name in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json"):  # the newest committed pass of this workload
        pmc_file = os.path.join(ROOT, "profiles", pmc_name)
        if os.path.exists(pmc_file) and a.pairs == 32 and a.width == 640 and a.levels == 4:
            try:
                pmc = json.load(open(pmc_file))
                if pmc.get("kernel_sha16") != src_now:
                    traffic_src = ("none: profiles/%s was taken with kernel sources %s, this run has %s -- re-run the PMC passes "
                                   "(profiles/README.md)" % (pmc_name, pmc.get("kernel_sha16", "(unstamped)"), src_now))
                    continue
                traffic = float(next(v for k, v in pmc.items() if k.startswith("k_track"))["hbm_bytes_per_launch"])
                traffic_commit = pmc.get("commit")
                traffic_src = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of the default pipelined "
                               "command; a committed measurement of the builder's, not taken inside this run; same kernel "
                               "sources: kernel_sha16 %s)" % (pmc_name, src_now))
                break
            except Exception:
                traffic = None

    # ---- per-kernel roofline table of the build (VERDICT r04 #6b): every kernel alone (HIP events between the launches,
    # revo_batch_profile_build), its algorithmic bytes per launch (what the kernel must read and write once,
    # 2*pairs frames per launch, `pairs` keyframes for the EDT) and the fraction of the HBM peak that makes
    P = [(a.width >> l) * (a.height >> l) for l in range(a.levels)]
    SP = float(sum(P))
    nframes = 2 * a.pairs
    has_orig = [l >= 1 and hist[l] > 0 and hist[l - 1] > 0 for l in range(a.levels)]
    P_orig = float(sum(P[l] for l in range(a.levels) if has_orig[l]))
    SN = float(npts.sum(1).mean())  # edge points per frame, all levels (the current frames of the batches)
    kbytes = {"k_gray_depth": nframes * 4.0 * P[0],
              "k_canny_nms4": nframes * (SP + SP / 4),
              "hysteresis": nframes * (SP / 4 + SP + P_orig + SP / 8),
              "k_fill": None,
              "k_edge_prefix": nframes * ((SP - P[-1]) / 8 + (SP - P[-1]) / 16),  # edge bitmaps in, one 16-bit prefix per 32-pixel word out
              "k_tile_count": nframes * (SP / 8 + SP / 8),
              "k_pts_tiles": nframes * (SP / 8 + SP / 8 + 4 * SN + 16 * SN),
              "k_edt_cols": a.pairs * (SP / 8 + 2 * SP),
              "k_edt_rows": a.pairs * (2 * SP + 4 * SP)}
    for l in range(1, a.levels):  # the two halves of cv::pyrDown + FilterSubsampleWithHoles (launched apart since round 5)
        kbytes["k_pyrdown_gray[%d]" % l] = nframes * (1.0 * P[l - 1] + 1.0 * P[l])
        kbytes["k_pyrdown_depth[%d]" % l] = nframes * (4.0 * P[l - 1] + 4.0 * P[l] + P[l - 1] / 8)
    kernels = []
    try:
        for name, us in bt.profile_build(d_bgr.data_ptr(), d_dep.data_ptr(), reps=3):
            kb = kbytes.get(name)
            kernels.append({"kernel": name, "us_alone": us, "algorithmic_bytes_per_launch": kb,
                            "frac": (kb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS) if (kb and us > 0) else None})
    except Exception as e:  # noqa: BLE001 -- a side table must not cost the line
        kernels = [{"error": "%s: %s" % (type(e).__name__, e)}]
    kernels.append({"kernel": "k_track", "us_alone": ms_track_alone * 1e3, "us_in_step": ms_track * 1e3,
                    "algorithmic_bytes_per_launch": b_trk, "frac": b_trk / (ms_track_alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "frac_in_step": achieved / HBM_PEAK_GBS})
    # the whole step against the roofline, SURVEY 8(d): per independent frame-pair 2*B_pyr + B_kf + B_trk
    b_pyr = 7.0 * P[0] + SP + 4.0 * (SP - P[0]) + SP + P_orig + 16.0 * SN
    b_kf = SP + 4.0 * SP + 16.0 * SP
    step_bytes_alg = a.pairs * (2 * b_pyr + b_kf) + b_trk
    # on-box streaming ceiling next to the vendor peak: a 1 GiB device-to-device copy
    copy_gbs = None
    if rank == 0:
        try:
            src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
        except RuntimeError:
            copy_gbs = None
    seen, group_size = parallel.ranks_seen(world, device=ctl_dev) if use_group else ([0], 1)
    if use_group and (len(set(seen)) != world or group_size != world):
        raise SystemExit("bench: the process group saw ranks %s (size %d) but WORLD_SIZE is %d" % (seen, group_size, world))
    gather_us = None
    if use_group:  # the collective alone (latency-bound: 96 B x pairs per rank)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d_alone = torch.zeros(world * a.pairs * parallel.RECORD_BYTES, dtype=torch.uint8, device=dev)
        e0.record()
        for _ in range(20):
            if native:
                comm.allgather_records(d_res.data_ptr(), d_alone.data_ptr(), a.pairs, stream=torch.cuda.current_stream().cuda_stream)
            else:
                parallel.gather_records(d_res, world, out=d_alone)
        e1.record()
        torch.cuda.synchronize()
        gather_us = e0.elapsed_time(e1) * 1e3 / 20
        if native and not torch.equal(d_alone[rank * a.pairs * 96:(rank + 1) * a.pairs * 96], d_res):
            raise SystemExit("bench: revo_comm_allgather_records did not return this rank's records")
    out = {
        "metric": "tracked frames/sec at 640x480, 4-level pyramid; ATE vs reference",
        "value": world * a.pairs * a.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "ms_per_step_per_rank": ms_per_rank,  # every rank's own clock around the same timed region (a straggler GPU shows here)
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        # the same step with fewer batches resident (this rank, no collective): ONE batch and nothing overlapped / TWO batches
        "value_single_batch_in_flight": value_single * world if value_single else None,
        "value_two_batches": value_two * world if value_two else None,
        "value_pairs_resident": {"value": nbuf * a.pairs, "value_two_batches": 2 * a.pairs, "value_single_batch_in_flight": a.pairs},
        "config": {
            "workload": "synthetic %dx%d RGB-D, %d-level pyramid, %d independent frame-pairs per step and GPU "
                        "(%s); per pair: 2 pyramid builds + keyframe (EDT+table) + trackFrames"
                        % (a.width, a.height, a.levels, a.pairs, baseline_config(a.width, a.height, a.levels, a.pairs, world)),
            "pairs_per_gpu": a.pairs, "global_pairs": world * a.pairs,
            "parallelism": "pairs sharded over %d GPU(s), one RCCL all_gather of 96 B/pair x %d step(s) every %d step(s)%s" % (world, every, every, " (native: revo_pipeline_set_comm)" if native else ""),
            "pipeline": {"owner": "library (revo_pipeline_*)", **{k: v for k, v in pipe_info.items() if k != "streams"}},
            "pipelining": "none" if a.no_overlap else ("%d batches of %d pairs in rotation: the build of a later step overlaps the tracker grids of "
                                                        "earlier ones; consecutive tracker grids on %d stream(s), ordered by the library's "
                                                        "resident gate (at most %s in flight)%s" % (nbuf, a.pairs, len(s_tracks), os.environ.get("REVO_TRACK_DEPTH", "2"),
                                                                                                  "; what the build leaves to its first consumer (REVO_DEFER=%s: 1 = the keyframes' EDT, 2 = + the edge lists, 3 = + hysteresis) on %d stream(s) of its own; the RCCL gather %s" % (os.environ.get("REVO_DEFER", "2"), len(s_edts), "on the tracker's stream, behind the grid") if s_edts else "")),
            "batches_in_rotation": nbuf, "pairs_resident": nbuf * a.pairs,
        },
        "roofline": {
            "bound": "hbm", "kernel": "k_track", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": traffic_src, "traffic_commit": traffic_commit, "kernel_sha16": src_now,
            "algorithmic_bytes_per_launch": b_trk, "kernel_ms": ms_track, "kernel_ms_alone": ms_track_alone,
            "frac_alone": b_trk / (ms_track_alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # two tracker grids are in flight (resident gate): a launch lasts longer than the interval at which launches
            # complete; `frac` above is per launch as the rules ask, this is the same bytes over the step interval
            "frac_per_step_interval": b_trk / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS,
            "timing": "HIP events on the tracker's stream around %d of the %d launches of the timed region (every %d-th%s)"
                      % (max(1, n_timed), a.steps, TIME_EVERY, "; recorded by the pipeline handle inside the tracker chain, directly around the grid: revo_pipeline_time_tracker"),
            "measured_copy_gbs": copy_gbs,  # on-box device-to-device copy ceiling (read + write bytes), for context
            # the WHOLE step against the roofline: SURVEY 8(d)'s bytes per independent frame-pair (2 B_pyr + B_kf + B_trk, with
            # this run's point counts and evaluation counts) x pairs / ms_per_step / peak -- per GPU
            "step": {"algorithmic_bytes_per_step": step_bytes_alg, "b_pyr_per_frame": b_pyr, "b_kf_per_keyframe": b_kf,
                     "b_trk_per_pair": b_trk / a.pairs, "achieved": step_bytes_alg / (elapsed / a.steps) / 1e9,
                     "frac": step_bytes_alg / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS},
            # every kernel of the step ALONE (build kernels: HIP events between the launches, one stream; k_track: kernel_ms_alone):
            # the kernel furthest below its roofline is the one to look at next
            "kernels": kernels,
        },
        "collective": {"backend": ("RCCL through the library's own communicator (revo_comm_* / revo_pipeline_set_comm; %s, nccl %d); control plane: gloo"
                                   % api.rccl_available() if native else "nccl (RCCL)") if use_group else None,
                       "enqueued_by": ("librevo_hip.so (pipeline handle, after-grid slot)" if native else "bench.py (torch.distributed, after-grid slot)") if use_group else None,
                       "executed_every_step": bool(use_group) and every == 1,
                       "steps_per_collective": every if use_group else None, "inside_timed_region": bool(use_group),
                       "bytes_per_rank": a.pairs * parallel.RECORD_BYTES * every, "us_per_all_gather_alone": gather_us,
                       "ranks_seen": seen, "world_size": group_size,  # all_gather of the rank ids / dist.get_world_size()
                       "error": group_error},
        "stages_ms": {"pyramids_and_keyframes": ms_build, "tracker": ms_trk_stage},
        "resident_gate": gate,  # after the timed region: time-outs must be 0
        "pose_error_vs_ground_truth": {"rot_rad_median": rot_med, "trans_m_median": tr_med},
        "mean_edge_points_lvl0": float(npts[:, 0].mean()),
        "mean_evals_per_level": [float(x) for x in evals.mean(0)],
        "evals_raw_mean": [float(x) for x in np.concatenate(evals_all).mean(0)],
        "inputs": {"batches_in_rotation": nin, "bytes_per_batch": int(bgrs[0].nbytes + deps[0].nbytes),
                   "note": "step t reads input batch t mod %d (distinct seeds); together they exceed the 256 MB Infinity Cache" % nin},
        "input_render_s": t_render,
    }

    if tum_out is not None:
        out["tum_stream"] = tum_out

    # the collective has done its job (and been timed): RCCL's proxy threads must not compete with the host-side
    # measurements below (IO thread + consumer thread of the sequential stream) for the cgroup's CPUs
    torch.cuda.synchronize()
    if native:
        pipe.set_comm(None, 0, None, 0)
        comm.close()
    if dist.is_initialized():
        dist.destroy_process_group()

    # ---- the same step fed from HOST memory (revo_track_pairs_submit / _wait): page-locked frames as a decoder
    # thread would leave them, H2D on its own stream overlapped with the kernels of the previous job.  Reported
    # next to `value` (which has its inputs resident in HBM); PCIe-bound.
    if rank == 0 and world == 1 and not a.skip_host_buffers:
        from collections import deque
        hostb = {}
        for tag, scale in (("u16", 5000.0), ("f32", None)):
            # the frames of a job lie back to back in TWO page-locked slabs, one per plane type ([2n][H][W][3] colours, [2n][H][W]
            # depths): what a decoder that fills job-sized staging leaves behind -- the library then moves each slab in one
            # copy (round 6; separately pinned frames, one copy each, reached 40 GB/s: REVO_BENCH_HOST_SLABS=0 restores them)
            slabs = os.environ.get("REVO_BENCH_HOST_SLABS", "1") != "0"
            nfr = 2 * len(rendered)
            slab_c = torch.empty((nfr, a.height, a.width, 3), dtype=torch.uint8).pin_memory().numpy()
            slab_d = torch.empty((nfr, a.height, a.width), dtype=torch.int16 if scale else torch.float32).pin_memory().numpy()
            if scale:
                slab_d = slab_d.view(np.uint16)
            fr = []
            for i, r in enumerate(rendered):
                one = []
                for k, (kb, kd) in enumerate(((0, 1), (2, 3))):
                    d = np.clip(r[kd] * 5000.0, 0, 65535).astype(np.uint16) if scale else r[kd]
                    if slabs:
                        slab_c[2 * i + k] = r[kb]
                        slab_d[2 * i + k] = d
                        one.append((slab_c[2 * i + k], slab_d[2 * i + k]))
                    else:
                        one.append((torch.from_numpy(np.ascontiguousarray(r[kb])).pin_memory().numpy(),
                                    torch.from_numpy(np.ascontiguousarray(d)).pin_memory().numpy()))
                fr.append(tuple(one))
            hb = api.HostBatchTracker(cam, depth_scale_factor=scale)
            ref_res = hb.track(fr)  # the records every later job must repeat
            for j in [hb.submit(fr) for _ in range(3)]:  # warm-up: all three job slots exist before the clock starts
                hb.wait(j)
            k_steps = max(8, a.steps)
            reps_h = []
            # FOUR repetitions of the same pipelined loop; the first is reported but not part of the statistic (VERDICT r04 #9/#6d):
            # in every run of rounds 3-5 the first repetition of the f32 leg moved 16-28 GB/s and the following ones 45, whatever
            # came before it (three single jobs, or untimed groups of three jobs: round 5, profiles/r05_call6.sh) -- the loop has
            # to have run once in its pipelined shape; the median of the three repetitions after it is the figure
            for rep in range(4):
                jobs = deque()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(k_steps):
                    jobs.append(hb.submit(fr))
                    if len(jobs) == 3:
                        last = hb.wait(jobs.popleft())
                while jobs:
                    last = hb.wait(jobs.popleft())
                reps_h.append(time.perf_counter() - t0)
                if any(not (np.array_equal(x["R"], y["R"]) and np.array_equal(x["T"], y["T"])) for x, y in zip(last, ref_res)):
                    raise SystemExit("bench: pipelined host-buffer jobs disagree with the first one")
            first_rep, reps_h = reps_h[0], reps_h[1:]
            dt_h = float(np.median(reps_h))
            step_bytes = a.pairs * 2 * a.width * a.height * (3 + (2 if scale else 4))
            hostb[tag] = {"value_incl_h2d": a.pairs * k_steps / dt_h, "unit": "frames/s", "ms_per_step": dt_h / k_steps * 1e3,
                          "h2d_bytes_per_step": step_bytes, "pcie_gbs": step_bytes * k_steps / dt_h / 1e9, "steps": k_steps,
                          "statistic": "median of the 3 repetitions after one untimed-in-the-statistic repetition of the same loop",
                          "host_layout": "one page-locked slab per plane type and job (one copy each)" if slabs else "one page-locked buffer per frame and plane",
                          "value_incl_h2d_first_repetition": a.pairs * k_steps / first_rep,
                          "value_incl_h2d_runs": [a.pairs * k_steps / t for t in reps_h],
                          "pcie_gbs_runs": [step_bytes * k_steps / t / 1e9 for t in reps_h]}
            del hb
        hostb["note"] = ("revo_track_pairs_submit/_wait from page-locked host frames, 3 jobs in flight: the H2D of job k+1 overlaps the "
                         "kernels of job k; u16 = raw depth as on disk (conversion fused into the build), f32 = metres")
        out["host_buffers"] = hostb
        out["value_incl_h2d"] = hostb["u16"]["value_incl_h2d"]

    # ---- the sequential stream measured at the start: its CPU counterpart and the record
    if seq_gpu is not None:
        n = n_seq
        dt_seq = float(np.median(seq_gpu["runs"]))
        runs = seq_gpu["runs"]
        ate_seq = seq_gpu["ate"]
        rpe_t, rpe_r = seq_gpu["rpe"]
        cpu_seq = cpu_seq_2core = cpu_trk_only = None
        traj_rmse = ate_gpu_n = ate_cpu_n = None
        nseq = 0
        seq_pinned, seq_passes = None, 0
        if a.cpu_baseline != "off":  # the same stream through the oracle's REVO::start restatement
            from oracle import ro
            nseq = min(n, 40)
            ovo = ro.VO(s)
            t0 = time.perf_counter()
            o_poses = [ovo.push(*fr)[0] for fr in stream_frames[:nseq]]
            cpu_seq = nseq / (time.perf_counter() - t0)  # everything on one core
            # "ATE vs reference" (the metric's second half): the device trajectory against the oracle's on the same frames
            traj_rmse = synth.ate_rmse(seq_gpu["poses"][:nseq], o_poses)
            ate_gpu_n = synth.ate_rmse(seq_gpu["poses"][:nseq], seq_gpu["gt"][:nseq])
            ate_cpu_n = synth.ate_rmse(o_poses, seq_gpu["gt"][:nseq])
            t_pyr, t_kf, t_trk = ovo.times()  # seconds in: pyramid builds, makeKeyframe, tracking + vote
            cpu_trk_only = nseq / t_trk
            # the reference's threading (system.cpp:96): IO thread builds pyramids, main thread tracks -- measured,
            # each thread pinned to its own core, 1 warm-up + 5 passes, median
            _, allowed = cpu_info()
            seq_pinned = [allowed[0], allowed[1 % len(allowed)]]
            sb = np.stack([f[0] for f in stream_frames[:nseq]])
            sd = np.stack([f[1] for f in stream_frames[:nseq]])
            st = [f[2] for f in stream_frames[:nseq]]
            times_seq = []
            for _ in range(6):
                times_seq.append(ro.VO(s).run_pipelined(sb, sd, st, seq_pinned[0], seq_pinned[1])[0])
            seq_passes = len(times_seq) - 1
            cpu_seq_2core = nseq / float(np.median(times_seq[1:]))
        out["single_stream"] = {"frames_per_s": n / dt_seq, "statistic": "median of %d runs" % len(runs),
                                "frames_per_s_runs": [n / t for t in runs], "frames": n,
                                "io_thread_longest_section_ms": seq_gpu["io_max_ms"],
                                "keyframes": seq_gpu["keyframes"],
                                "ate_rmse_vs_ground_truth_m": ate_seq,
                                # the oracle ran the same frames (its first `oracle_frames`): trajectory against trajectory, and
                                # the difference of the two ATEs against the synthetic ground truth (north star: within 1 mm)
                                "trajectory_rmse_gpu_vs_oracle_m": traj_rmse,
                                "ate_difference_vs_oracle_m": (abs(ate_gpu_n - ate_cpu_n) if traj_rmse is not None else None),
                                "ate_rmse_oracle_vs_ground_truth_m": ate_cpu_n, "oracle_frames": nseq,
                                "rpe_rmse_per_frame": {"trans_m": rpe_t, "rot_rad": rpe_r},
                                "cpu_oracle_frames_per_s_1core": cpu_seq,
                                "cpu_oracle_frames_per_s_2core_pipelined": cpu_seq_2core,
                                "cpu_oracle_2core": {"pinned_cpus": seq_pinned, "passes": seq_passes, "statistic": "median",
                                                     "threads": "IO thread (pyramids) + main thread (keyframes, tracking, vote)"},
                                "cpu_oracle_tracker_only_frames_per_s_1core": cpu_trk_only if cpu_seq else None,
                                "speedup_vs_cpu_oracle_2core_pipelined": (n / dt_seq / cpu_seq_2core) if cpu_seq else None,
                                "speedup_vs_cpu_oracle": (n / dt_seq / cpu_seq) if cpu_seq else None,
                                "note": "sequential REVO::start sequencing (revo_vo_*) via the host-buffer C ABI on a seeded synthetic "
                                        "sweep, frames in page-locked host memory: H2D (in place, no staging copy) + pyramid on the IO "
                                        "thread, trackFrames + quality vote per frame on the consumer thread (PCIe-inclusive)"}

    # ---- CPU baseline: the oracle (plain-C port of the reference's algorithm) on the same batch
    # with the reference's thread model -- the IO thread builds the pyramids, the main thread promotes keyframes and
    # tracks (system.cpp:96) -- each pinned to its own core, 1 warm-up + 5 passes, median.  Bounded: ~0.3 s per pass.
    if rank == 0 and a.cpu_baseline != "off":  # at every N: measured once, on rank 0 (its input batch 0), after the timed region
        from oracle import ro
        model, allowed = cpu_info()
        pinned = [allowed[0], allowed[1 % len(allowed)]]
        passes = ro.bench_pairs_pipelined(s, bgr, dep, pinned[0], pinned[1], 6)
        med = float(np.median(passes[1:]))
        out["cpu_baseline"] = {
            "value": a.pairs / med, "unit": "frames/s", "cores": 2, "kind": "port",
            "pinned_cpus": pinned, "passes": len(passes) - 1, "warmup_passes": 1, "statistic": "median",
            "pass_seconds": passes[1:], "cpu_model": model, "nproc": os.cpu_count(), "cpus_allowed": len(allowed),
            "cpus_granted_by_cgroup": usable_cpus(),
            "sample": "the %d frame-pairs of the timed batch per pass (2 pyramids + keyframe + trackFrames each), oracle/ "
                      "plain-C restatement, gcc -O3 -mavx2, IO thread + tracker thread like the reference" % a.pairs,
        }
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0 and world == 1 and a.cpu_baseline != "off":
        # one core (everything on the tracker's thread), same pinning, 1 warm-up + 3 passes
        trk = ro.Tracker(s)
        I3, Z3 = np.eye(3), np.zeros(3)
        os.sched_setaffinity(0, {pinned[1]})
        try:
            one = []
            for _ in range(4):
                t0 = time.perf_counter()
                for i in range(a.pairs):
                    o_ref = ro.Pyramid(s, rendered[i][0], rendered[i][1])
                    o_cur = ro.Pyramid(s, rendered[i][2], rendered[i][3])
                    o_ref.makeKeyframe()
                    trk.trackFrames(o_ref, o_cur, I3, Z3)
                one.append(time.perf_counter() - t0)
        finally:
            os.sched_setaffinity(0, set(allowed))
        out["cpu_baseline_1core"] = {"value": a.pairs / float(np.median(one[1:])), "unit": "frames/s", "cores": 1, "kind": "port",
                                     "pinned_cpus": [pinned[1]], "passes": len(one) - 1, "statistic": "median"}
        out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline_1core"]["value"]
        # context: the same port on ALL granted hardware threads, one pair per
        # thread at a time (native pthreads inside the oracle library), bounded to a few seconds
        ncore = usable_cpus()
        total, all_t = ro.bench_pairs_mt(s, bgr, dep, ncore, min(4.0, a.cpu_seconds))
        out["cpu_baseline_all_cores"] = {
            "value": total / all_t, "unit": "frames/s", "cores": ncore, "kind": "port",
            "sample": "%d frame-pairs over %d threads (= CPUs granted by the cgroup quota) in %.1f s, native pthreads, "
                      "same per-pair work as cpu_baseline" % (total, ncore, all_t)}
        out["speedup_vs_cpu_all_cores"] = out["value"] / out["cpu_baseline_all_cores"]["value"]
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
