"""What an occasional SLOW run of the sequential stream coincides with ("the 1-in-25 slow run explained or gone").

  python profiles/slow_run_probe.py [runs=80] [frames=60] [pin]           (GPU box)

Runs the bench's 60-frame sweep `runs` times through the IO-thread driver with pauses of 0 / 2 / 20 / 200 ms in front of a run
(in rotation) while a SEPARATE process samples the graphics clock, the activity and the power through amdsmi (a thread of this
process would fight the consumer loop for the interpreter lock -- and make slow runs of its own).  Per run: frames/s, the pause
before it, the clock samples inside its window (min / mean MHz), the cgroup's CPU throttling (cpu.stat nr_throttled /
throttled_usec) and the process's involuntary context switches across it.  Prints every run, then the slow ones (below 85 % of
the median) with everything known about them, incl. the largest gaps between consecutive poses and where in the run they fell.
`pin`: the frames sit in page-locked memory (torch.pin_memory, as bench.py has them) and are read in place by the DMA engine.
"""
import multiprocessing as mp
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sampler(stop, path):
    rows = []
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]
    except Exception as e:  # noqa: BLE001
        np.save(path, np.zeros((0, 4)))
        print("sampler: amdsmi unavailable: %r" % (e,), flush=True)
        return

    def clk():
        try:
            c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
            return float(c.get("clk", c.get("cur_clk", -1)))
        except Exception:  # noqa: BLE001
            return -1.0

    def act_pow():
        a = p = -1.0
        try:
            a = float(amdsmi.amdsmi_get_gpu_activity(h).get("gfx_activity", -1))
        except Exception:  # noqa: BLE001
            pass
        try:
            pi = amdsmi.amdsmi_get_power_info(h)
            v = pi.get("current_socket_power", pi.get("average_socket_power", -1))
            p = float(v) if not isinstance(v, str) else -1.0
        except Exception:  # noqa: BLE001
            pass
        return a, p

    k = 0
    a = p = -1.0
    while not stop.is_set():
        t = time.perf_counter()
        c = clk()
        if k % 8 == 0:
            a, p = act_pow()
        rows.append((t, c, a, p))
        k += 1
    np.save(path, np.array(rows, np.float64))


def cpu_stat():
    out = {}
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for ln in open(f):
                k, v = ln.split()
                out[k] = int(v)
            break
        except OSError:
            continue
    return out.get("nr_throttled", -1), out.get("throttled_usec", out.get("throttled_time", -1))


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    from revo_amd import api, synth, vo
    from revo_amd.settings import ImgPyramidSettings
    s = ImgPyramidSettings(pyr_min_lvl=3)
    seq = synth.make_sequence(7, s, n, max_t=0.01, max_rot_deg=0.4, bias=[0.004, 0, 0, 0, np.deg2rad(1.0), 0])
    frames = [f[:3] for f in seq]
    pin = len(sys.argv) > 3 and sys.argv[3] == "pin"
    if pin:
        import torch
        frames = [(torch.from_numpy(np.ascontiguousarray(f[0])).pin_memory().numpy(),
                   torch.from_numpy(np.ascontiguousarray(f[1])).pin_memory().numpy(), f[2]) for f in frames]
    if os.environ.get("PROBE_SCHEDULE"):  # hipSetDeviceFlags: 1 = hipDeviceScheduleSpin, 2 = Yield, 4 = BlockingSync
        import ctypes as C0
        hip = C0.CDLL("libamdhip64.so")
        print("hipSetDeviceFlags(%s) -> %d" % (os.environ["PROBE_SCHEDULE"], hip.hipSetDeviceFlags(int(os.environ["PROBE_SCHEDULE"]))))
    cam = api.CameraPyr(s)
    for _ in range(2):
        vo.REVO(s, cameraPyr=cam).run(frames)
    ctx = mp.get_context("spawn")
    stop = ctx.Event()
    path = "/tmp/slow_run_probe_samples.npy"
    proc = ctx.Process(target=sampler, args=(stop, path))
    proc.start()
    time.sleep(1.0)
    pauses = [0.0, 0.002, 0.02, 0.2]
    rec = []
    all_stamps = []
    all_sub = []
    import gc
    for r in range(runs):
        pause = pauses[r % len(pauses)]
        if pause:
            time.sleep(pause)
        drv = vo.REVO(s, cameraPyr=cam)
        stamps = []
        inner = drv.track_next

        def stamped(inner=inner, stamps=stamps):
            r = inner()
            stamps.append(time.perf_counter())
            return r
        drv.track_next = stamped
        sub = []
        inner_submit = drv.submit

        def submit_stamped(*a, inner_submit=inner_submit, sub=sub):
            ta = time.perf_counter()
            inner_submit(*a)
            sub.append((ta, time.perf_counter()))
        drv.submit = submit_stamped
        gc.collect()
        gc.disable()
        thr0, ru0 = cpu_stat(), resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        drv.run(frames)
        t1 = time.perf_counter()
        gc.enable()
        all_stamps.append(np.array([t0] + stamps))
        all_sub.append(np.array(sub))
        thr1, ru1 = cpu_stat(), resource.getrusage(resource.RUSAGE_SELF)
        rec.append((t0, t1, pause, thr1[0] - thr0[0], thr1[1] - thr0[1], ru1.ru_nivcsw - ru0.ru_nivcsw, drv.nKeyFrames))
        del drv
    stop.set()
    proc.join(20)
    smp = np.load(path) if os.path.exists(path) else np.zeros((0, 4))
    fps = np.array([n / (b - a) for a, b, *_ in rec])
    med = float(np.median(fps))
    print("%d runs of %d frames: median %.0f frames/s, min %.0f, max %.0f; %d clock samples (%.2f ms apart)"
          % (runs, n, med, fps.min(), fps.max(), len(smp), 1e3 * float(np.median(np.diff(smp[:, 0]))) if len(smp) > 1 else -1))
    if len(smp):
        print("graphics clock over the whole probe: min %.0f MHz, median %.0f, max %.0f; activity median %.0f %%; power median %.0f W"
              % (smp[:, 1].min(), np.median(smp[:, 1]), smp[:, 1].max(), np.median(smp[:, 2]), np.median(smp[:, 3])))
    print("%4s %9s %8s %10s %10s %10s %9s %9s %8s" % ("run", "frames/s", "pause ms", "clk min", "clk mean", "clk before", "throttled", "thr usec", "invol cs"))
    lines = []
    for i, (a, b, pause, dthr, dus, dcs, kf) in enumerate(rec):
        inside = smp[(smp[:, 0] >= a) & (smp[:, 0] <= b)] if len(smp) else smp
        before = smp[(smp[:, 0] >= a - 0.003) & (smp[:, 0] < a)] if len(smp) else smp
        cmin = inside[:, 1].min() if len(inside) else -1
        cmean = inside[:, 1].mean() if len(inside) else -1
        cbef = before[:, 1].mean() if len(before) else -1
        ln = "%4d %9.0f %8.1f %10.0f %10.0f %10.0f %9d %9d %8d" % (i, fps[i], 1e3 * pause, cmin, cmean, cbef, dthr, dus, dcs)
        lines.append(ln)
        print(ln)
    slow = [i for i in range(runs) if fps[i] < 0.85 * med]
    print("slow runs (< 85 %% of the median): %d of %d" % (len(slow), runs))
    for i in slow:
        print(lines[i])
        d = np.diff(all_stamps[i]) * 1e3
        top = np.argsort(d)[::-1][:4]
        print("      largest gaps between poses (ms @ frame): " + ", ".join("%.2f @ %d" % (d[k], k) for k in top)
              + "; median gap %.3f ms" % float(np.median(d)))
        sd = (all_sub[i][:, 1] - all_sub[i][:, 0]) * 1e3
        top = np.argsort(sd)[::-1][:3]
        k = int(np.argmax(d))
        print("      longest submit calls of the IO thread (ms @ frame, started ms before the late pose): "
              + ", ".join("%.2f @ %d (%.2f)" % (sd[j], j, 1e3 * (all_stamps[i][k + 1] - all_sub[i][j, 0])) for j in top)
              + "; median submit %.3f ms" % float(np.median(sd)))
    d_all = np.concatenate([np.diff(x) for x in all_stamps]) * 1e3
    print("gaps between consecutive poses over all runs: median %.3f ms, 99 %% %.3f, 99.9 %% %.3f, max %.3f; gaps > 1 ms: %d of %d"
          % (np.median(d_all), np.percentile(d_all, 99), np.percentile(d_all, 99.9), d_all.max(), int((d_all > 1.0).sum()), len(d_all)))
    try:
        import ctypes as C
        from revo_amd import _lib
        st = (C.c_ulonglong * 3)()
        _lib.lib().revo_debug_wait_stats_(st)
        print("host waits for a result word: %d calls, %d ran out of their spin budget (then hipStreamSynchronize), longest wait %.3f ms"
              % (st[0], st[1], st[2] / 1e6))
    except Exception as e:  # noqa: BLE001
        print("no wait statistics: %r" % (e,))
    try:
        sec = (C.c_ulonglong * 12)()
        _lib.lib().revo_debug_section_max_(sec, 0)
        names = ["lock + pool", "pointer attributes", "build stream waits for copies", "wait for copies", "enqueue build", "wait for queue room",
                 "copy stream waits", "hipMemcpyAsync colour", "hipMemcpyAsync depth", "hipEventRecord"]
        print("longest a section of a frame submission ever took (ms): " + ", ".join("%s %.3f" % (nm, sec[i] / 1e6) for i, nm in enumerate(names)))
    except Exception as e:  # noqa: BLE001
        print("no section statistics: %r" % (e,))
    by = {}
    for i, r in enumerate(rec):
        by.setdefault(r[2], []).append(fps[i])
    for pause in sorted(by):
        v = np.array(by[pause])
        print("pause %5.1f ms in front: median %.0f frames/s, min %.0f, %d of %d slow" % (1e3 * pause, np.median(v), v.min(), int((v < 0.85 * med).sum()), len(v)))


if __name__ == "__main__":
    main()
