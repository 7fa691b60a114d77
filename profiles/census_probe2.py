"""Diagnostic 2: the bench flow (VO leg -> RCCL init -> pipeline -> steps with the gather in the after-grid slot) with the
census printed after every stage and every step."""
import ctypes as C
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
import torch
import torch.distributed as dist
from revo_amd import api, synth, vo, _lib, parallel
from revo_amd.settings import ImgPyramidSettings, TrackerSettings

def main():
    L = _lib.lib()
    L.revo_debug_census_.argtypes = [C.c_int, C.POINTER(C.c_uint)]
    sync_each = "nosync" not in sys.argv
    use_vo = "novo" not in sys.argv
    use_coll = "nocoll" not in sys.argv


    def census(tag, sync=True):
        if sync:
            torch.cuda.synchronize()
        o = (C.c_uint * 3)()
        rc = L.revo_debug_census_(0, o)
        print("%-44s census %d timeouts %d enqueued %d  diff %d" % (tag, o[0], o[1], o[2], o[2] - o[0]), flush=True)


    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    cam = api.CameraPyr(s, device=0)
    api.TrackerNew(TrackerSettings(), s, cam)
    if use_vo:
        seq = synth.make_sequence(7, s, 20, max_t=0.01, max_rot_deg=0.4, bias=[0.004, 0, 0, 0, np.deg2rad(1.0), 0])
        pin = lambda x: torch.from_numpy(np.ascontiguousarray(x)).pin_memory().numpy()
        frames = [(pin(f[0]), pin(f[1]), f[2]) for f in seq]
        for r in range(2):
            d = vo.REVO(s, cameraPyr=cam)
            d.run(frames)
            census("after VO run %d" % r)
            del d
    if use_coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(parallel.free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        census("after RCCL init")
    n = 8
    pairs = [synth.make_pair(i, s) for i in range(n)]
    bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).to(dev)
    dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).to(dev)
    pipe = api.Pipeline(cam, n)
    print(pipe.info())
    census("after pipeline create")
    outs = [torch.zeros(n * 96, dtype=torch.uint8, device=dev) for _ in range(16)]
    d_all = [torch.zeros(n * 96, dtype=torch.uint8, device=dev) for _ in range(2)]
    ext = {}
    for t in range(16):
        t0 = time.perf_counter()
        ticket, sh = pipe.submit(bgr.data_ptr(), dep.data_ptr(), outs[t].data_ptr())
        if use_coll and t >= 6:  # RCCL is initialised from the start, its collectives start at step 6
            st = ext.setdefault(sh, torch.cuda.ExternalStream(sh, device=dev))
            with torch.cuda.stream(st):
                parallel.gather_records(outs[t], 1, out=d_all[t % 2])
        if sync_each or t == 15:
            census("step %d (%.1f ms)" % (t, (time.perf_counter() - t0) * 1e3), sync=True)
    res = api.results_from_buffer(outs[15].cpu().numpy().tobytes(), n)
    print("flags", sorted(set(r["flags"] for r in res)))


if __name__ == "__main__":
    main()
