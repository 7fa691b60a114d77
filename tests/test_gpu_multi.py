"""Multi-GPU path on real hardware: the world-size-1 RCCL group is the single-GPU CI of the
8-GPU job -- the same gather / max-time code that runs at N > 1 -- and `bench.py --gpus N` must really start
N ranks (a GPU-less rank fails loudly instead of the job silently running on one)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORLD1 = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from revo_amd import api, parallel, synth
from revo_amd.settings import ImgPyramidSettings, TrackerSettings
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
cam = api.CameraPyr(s, device=0)
api.TrackerNew(TrackerSettings(), s, cam)
n = 3
pairs = [synth.make_pair(40 + i, s) for i in parallel.shard_pairs(n, 0, 1)]
bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).to(dev)
dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).to(dev)
bt = api.BatchTracker(cam, n)
res = torch.zeros(n * parallel.RECORD_BYTES, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    bt.track(bgr.data_ptr(), dep.data_ptr(), res.data_ptr(), stream=st.cuda_stream)
    allrec = parallel.gather_records(res, 1)       # RCCL all_gather on the tracker's stream, world size 1
    t = parallel.max_over_ranks(1.25, 1, device=dev)  # RCCL all_reduce(MAX)
torch.cuda.synchronize()
assert allrec.data_ptr() != res.data_ptr() and torch.equal(allrec, res) and t == 1.25
R, T, err = parallel.records_to_poses(allrec.cpu().numpy().tobytes(), n)
recs = api.results_from_buffer(res.cpu().numpy().tobytes(), n)
for i, p in enumerate(pairs):
    assert recs[i]["flags"] & (2 | 4 | 8) == 0
    er, et = synth.pose_error(R[i], T[i], p["T_ref_curr"])
    assert er < 5e-3 and et < 5e-3, (i, er, et)
    assert np.array_equal(R[i], recs[i]["R"]) and np.array_equal(T[i], recs[i]["T"])
dist.destroy_process_group()
print("world1 rccl ok")
"""


def test_world_size_1_rccl_group_gathers_tracker_records():
    from revo_amd import parallel
    env = parallel.rendezvous_env(0, 1)
    r = subprocess.run([sys.executable, "-c", WORLD1 % ROOT], env=env, capture_output=True, timeout=600)
    assert r.returncode == 0 and b"world1 rccl ok" in r.stdout, (r.stdout.decode()[-2000:], r.stderr.decode()[-4000:])


def test_bench_gpus_n_starts_n_ranks():
    import torch
    ngpu = torch.cuda.device_count()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "2", "--steps", "2", "--warmup", "1",
           "--single-stream-frames", "0", "--cpu-baseline", "off", "--width", "320", "--height", "240", "--levels", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=900)
    if ngpu >= 2:
        assert r.returncode == 0, r.stderr.decode()[-4000:]
        line = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["config"]["global_pairs"] == 4 and line["collective"]["executed_every_step"]
    else:  # one GPU: rank 1 has no device -> the job fails loudly, it does not fall back to one rank
        assert r.returncode != 0
        assert b"needs GPU 1" in r.stderr or b"ranks failed" in r.stderr, r.stderr.decode()[-4000:]
