"""GPU: the pipelined batch mode behind ONE handle of the C ABI (revo_pipeline_*, include/revo_hip.h) -- the counterpart of
the reference's producer / consumer pipeline (system/system.cpp:96,128-284, io/iowrapperRGBD.cpp:279-288).  The handle owns
its four streams and batches; results must be the bits the same inputs produce through one batch alone on one stream."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from revo_amd import synth  # noqa: E402
from revo_amd.settings import ImgPyramidSettings, TrackerSettings, PLANE_DT  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from revo_amd import api as A
    return A


@pytest.fixture(scope="module")
def ro():
    from oracle import ro as R
    return R


def _inputs(s, n, nin, seed0, dev):
    import torch
    out = []
    for b in range(nin):
        pairs = [synth.make_pair(seed0 + 10 * b + i, s) for i in range(n)]
        bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).to(dev)
        dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).to(dev)
        out.append((bgr, dep))
    return out


def _alone(api, cam, n, inputs, dev):
    """the records every input batch produces through ONE batch on one stream"""
    import torch
    ref = []
    bt = api.BatchTracker(cam, n)
    for bgr, dep in inputs:
        res = torch.zeros(n * 96, dtype=torch.uint8, device=dev)
        bt.track(bgr.data_ptr(), dep.data_ptr(), res.data_ptr())
        bt.sync()
        ref.append(res.cpu().numpy().tobytes())
        assert all(r["flags"] & (2 | 4 | 8) == 0 for r in api.results_from_buffer(ref[-1], n))
    return ref


@pytest.mark.parametrize("depth", [4, 3, 2, 1])
def test_pipeline_handle_equals_the_single_batch_path(api, ro, depth):
    """Every step of every round through the library-owned pipeline (device records, after-grid slot used for a copy of the
    records on the returned stream) equals the batch alone, bit for bit; an accessor on the step's batch sees finished planes."""
    import torch
    dev = torch.device("cuda", 0)
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    n, nin = 8, 3
    inputs = _inputs(s, n, nin, 900, dev)
    ref = _alone(api, cam, n, inputs, dev)
    pipe = api.Pipeline(cam, n, depth=depth)
    info = pipe.info()
    assert info["batches"] == depth and info["pairs_per_step"] == n
    want_streams = 1 if depth == 1 else (3 if depth == 2 else 4)
    assert len(set(info["streams"])) == want_streams
    # the handle checked its streams: pairwise distinct hardware queues (HIP's default four suffice for four streams)
    assert info["distinct_hw_queues"] == want_streams, info
    steps = 5 * max(2, depth) + 1
    outs = [torch.zeros(n * 96, dtype=torch.uint8, device=dev) for _ in range(steps)]
    copies = [torch.zeros(n * 96, dtype=torch.uint8, device=dev) for _ in range(steps)]
    tickets = []
    for t in range(steps):
        bgr, dep = inputs[t % nin]
        ticket, sh = pipe.submit(bgr.data_ptr(), dep.data_ptr(), outs[t].data_ptr())
        assert ticket == t + 1 and sh in info["streams"][:2]
        with torch.cuda.stream(torch.cuda.ExternalStream(sh, device=dev)):  # the after-grid slot: runs behind the grid
            copies[t].copy_(outs[t], non_blocking=True)
        tickets.append(ticket)
    # the newest step's batch: an accessor sees finished lists and DT planes
    v = pipe.batch_frame(tickets[-1], 0, s)
    bgr, dep = inputs[(steps - 1) % nin]
    o = ro.Pyramid(s, bgr[0].cpu().numpy(), dep[0].cpu().numpy())
    o.makeKeyframe()
    assert np.array_equal(v.returnDistTransform(0), o.read(PLANE_DT, 0))
    pipe.wait(tickets[-1])   # covers the copy enqueued behind the last grid
    assert copies[-1].cpu().numpy().tobytes() == ref[(steps - 1) % nin]
    pipe.wait(tickets[0])    # a step whose slot has moved on: complete by construction
    pipe.drain()
    for t in range(steps):
        assert outs[t].cpu().numpy().tobytes() == ref[t % nin], "step %d differs from the batch alone (depth %d)" % (t, depth)
        assert copies[t].cpu().numpy().tobytes() == ref[t % nin], "after-grid copy of step %d ran before its grid" % t
    assert pipe.info()["steps_submitted"] == steps
    pipe.close()


def test_pipeline_host_results_and_back_pressure(api):
    """host_results: every step's records come back through pinned memory from revo_pipeline_wait; a slot whose step has not
    been waited for refuses the next submit (REVO_ERR_CAPACITY) instead of overwriting records nobody has read; u16 depth and
    initial poses go through the same entry."""
    import torch
    from revo_amd._lib import RevoError
    dev = torch.device("cuda", 0)
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    n, depth = 4, 3
    pairs = [synth.make_pair(40 + i, s) for i in range(n)]
    bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).to(dev)
    dep_f = np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])
    raw = np.clip(np.nan_to_num(dep_f, nan=0.0, posinf=0.0, neginf=0.0) * 5000.0, 0, 65535).astype(np.uint16)
    d_raw = torch.from_numpy(raw.view(np.int16)).to(dev)
    init = api.pack_init_RT([p["T_ref_curr"][:3, :3] for p in pairs], [p["T_ref_curr"][:3, 3] for p in pairs])
    # reference: one batch, u16 build, the same initial poses
    bt = api.BatchTracker(cam, n)
    res = torch.zeros(n * 96, dtype=torch.uint8, device=dev)
    bt.build_u16(bgr.data_ptr(), d_raw.data_ptr(), 5000.0)
    bt.track_only(res.data_ptr(), init_RT=init)
    bt.sync()
    want = api.results_from_buffer(res.cpu().numpy().tobytes(), n)
    pipe = api.Pipeline(cam, n, depth=depth, host_results=True)
    t = [pipe.submit(bgr.data_ptr(), d_raw.data_ptr(), None, init_RT=init, depth_kind=api.Pipeline.DEPTH_U16, depth_scale_factor=5000.0)[0]
         for _ in range(depth)]
    with pytest.raises(RevoError) as e:  # slot 0 still holds step 1's records
        pipe.submit(bgr.data_ptr(), d_raw.data_ptr(), None, init_RT=init, depth_kind=api.Pipeline.DEPTH_U16, depth_scale_factor=5000.0)
    assert e.value.code == -5
    for ticket in t:
        got = pipe.wait(ticket)
        for g, w in zip(got, want):
            assert np.array_equal(g["R"], w["R"]) and np.array_equal(g["T"], w["T"]) and g["evals"].tolist() == w["evals"].tolist()
            assert g["flags"] & (2 | 4 | 8) == 0
    t2, _ = pipe.submit(bgr.data_ptr(), d_raw.data_ptr(), None, init_RT=init, depth_kind=api.Pipeline.DEPTH_U16, depth_scale_factor=5000.0)
    assert t2 == depth + 1
    with pytest.raises(RevoError):  # step 1's records have been overwritten by step depth + 1
        pipe.wait(t[0])
    assert np.array_equal(pipe.wait(t2)[0]["R"], want[0]["R"])
    pipe.close()


def test_pipeline_live_tracker_timing(api):
    """revo_pipeline_time_tracker: HIP event pairs around the tracker grid on its own stream, inside the pipelined steps
    (what bench.py's roofline.kernel_ms is made of)."""
    import torch
    dev = torch.device("cuda", 0)
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    n = 8
    (bgr, dep), = _inputs(s, n, 1, 1200, dev)
    out = torch.zeros(n * 96, dtype=torch.uint8, device=dev)
    pipe = api.Pipeline(cam, n)
    pipe.time_tracker(2)
    for _ in range(12):
        pipe.submit(bgr.data_ptr(), dep.data_ptr(), out.data_ptr())
    pipe.drain()
    ms, launches = pipe.tracker_ms()
    assert launches == 6 and 0.01 < ms < 50.0, (ms, launches)
    pipe.close()


def test_cpp_host_drives_the_pipeline_through_the_c_abi(tmp_path):
    """tests/cpp/pipeline_host.cpp: a plain C++ host (no torch, no Python) builds against include/revo_hip.h, links
    librevo_hip.so and runs revo_pipeline_* on synthetic frames: records of the pipelined steps == the single-batch path."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "pipeline_host")
    so_dir = os.path.join(ROOT, "revo_amd")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "pipeline_host.cpp"), "-L", so_dir, "-lrevo_hip",
                           "-Wl,-rpath," + so_dir], timeout=300)
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    n = 6
    pairs = [synth.make_pair(1300 + i, s) for i in range(n)]
    np.stack([p[k][0] for p in pairs for k in ("ref", "curr")]).tofile(str(tmp_path / "bgr.bin"))
    np.stack([p[k][1] for p in pairs for k in ("ref", "curr")]).astype(np.float32).tofile(str(tmp_path / "depth.bin"))
    out = subprocess.run([exe, "320", "240", str(n), str(tmp_path / "bgr.bin"), str(tmp_path / "depth.bin")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "PIPELINE_HOST_OK" in out.stdout, out.stdout + out.stderr
    assert "4 distinct hardware queues" in out.stdout, out.stdout
    assert "native collective:" in out.stdout and "3 windows + 1 flushed step(s) gathered" in out.stdout, out.stdout


def test_pipeline_gathers_its_records_through_the_native_rccl_communicator(api):
    """SURVEY 8(e) behind the C ABI: revo_comm_* (RCCL loaded at run time, world size 1 here: the single-GPU CI of the 8-GPU
    job) + revo_pipeline_set_comm -- the handle enqueues one all-gather per window of `every` steps in the after-grid slot of
    the window's last step.  Every gathered record must be the bits the batch alone produces, for every = 1, 2, 3 and both
    pipeline depths that alternate tracker streams (4) and do not (2); revo_pipeline_flush_comm covers an incomplete window; a
    stand-alone revo_comm_allgather_records copies records bit for bit."""
    import torch
    from revo_amd._lib import RevoError
    dev = torch.device("cuda", 0)
    path, ver = api.rccl_available()
    assert ver > 0, (path, ver)
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    n, nin = 6, 3
    inputs = _inputs(s, n, nin, 2100, dev)
    ref = _alone(api, cam, n, inputs, dev)
    comm = api.Comm(cam, api.comm_unique_id(), 1, 0)
    # the collective alone
    send = torch.frombuffer(bytearray(ref[0]), dtype=torch.uint8).to(dev)
    recv = torch.zeros_like(send)
    comm.allgather_records(send.data_ptr(), recv.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert recv.cpu().numpy().tobytes() == ref[0]
    for depth, every in ((4, 2), (4, 1), (4, 3), (2, 2)):
        ring = 3
        pipe = api.Pipeline(cam, n, depth=depth)
        with pytest.raises(RevoError):  # no host results, no communicator, no buffer: the records would be unreachable
            pipe.submit(inputs[0][0].data_ptr(), inputs[0][1].data_ptr(), None)
        gathered = torch.zeros(ring * 1 * every * n * 96, dtype=torch.uint8, device=dev)
        pipe.set_comm(comm, every, gathered.data_ptr(), ring)
        steps = 4 * every + (1 if every > 1 else 0)
        own = torch.zeros(n * 96, dtype=torch.uint8, device=dev)
        for t in range(steps):
            bgr, dep = inputs[t % nin]
            ticket, _ = pipe.submit(bgr.data_ptr(), dep.data_ptr(), own.data_ptr() if t == 0 else None)
            if (t + 1) % every == 0:
                pipe.wait(ticket)
                k = t // every
                w = gathered.view(ring, every * n * 96)[k % ring].cpu().numpy().tobytes()
                for j in range(every):
                    assert w[j * n * 96:(j + 1) * n * 96] == ref[(k * every + j) % nin], (depth, every, t, j)
        valid, slot = pipe.flush_comm()
        pipe.drain()
        assert valid == steps % every and slot == (steps // every) % ring
        w = gathered.view(ring, every * n * 96)[slot].cpu().numpy().tobytes()
        for j in range(valid):
            assert w[j * n * 96:(j + 1) * n * 96] == ref[((steps // every) * every + j) % nin]
        assert own.cpu().numpy().tobytes() == ref[0]  # a caller buffer next to the communicator gets the step's own records
        pipe.set_comm(None, 0, None, 0)  # detach
        pipe.close()
    comm.close()
