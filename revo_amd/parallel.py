"""Multi-GPU layer of the batched mode: one process per GPU, frame-pairs
sharded statically across ranks, NO data-path collective; the only exchange is one
all_gather of the fixed 96-byte pair records (RCCL over xGMI on GPUs, gloo in the CPU
tests).  At 96 B x pairs the collective is latency-bound; nothing is reduced.

A process group of world size 1 is a real group: when one is initialised, the gather and
the max-time reduction go through the backend (RCCL on a GPU) exactly like at N > 1, so the
single-GPU CI exercises the same code path the 8-GPU job runs."""
import os
import socket
import subprocess
import sys

import numpy as np

RECORD_BYTES = 96


def shard_pairs(global_pairs, rank, world):
    """Static block partition: rank g owns pairs [g*B/G, (g+1)*B/G)."""
    if global_pairs % world:
        raise ValueError("global_pairs must be divisible by the world size")
    per = global_pairs // world
    return list(range(rank * per, (rank + 1) * per))


def _group_live():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def gather_records(local_records, world, group=None, out=None):
    """local_records: uint8 tensor [pairs*96] on this rank's device -> uint8 tensor
    [world*pairs*96] holding every rank's records in rank order (on every rank).
    `out`: optional preallocated result tensor (steady-state loops reuse one).
    Without a process group a single rank's records are their own gather."""
    import torch
    import torch.distributed as dist
    if world == 1 and not _group_live():
        return local_records
    if out is None:
        out = torch.empty(world * local_records.numel(), dtype=torch.uint8, device=local_records.device)
    elif out.numel() != world * local_records.numel() or out.dtype != torch.uint8:
        raise ValueError("out must be a uint8 tensor of world * len(local_records) elements")
    dist.all_gather_into_tensor(out, local_records, group=group)
    return out


def gather_every_default(world):
    """Steps per collective: 2 at every world size.  The collective sits in the after-grid slot of a tracker stream, in front of
    that stream's next grid, so every collective is a point where a rank that runs behind delays the others' next-but-one grid;
    carrying two steps' records per collective halves those points and lets a rank lag a full step in between (SURVEY 8e:
    latency-bound, no reduction, no ring).  The same schedule at N = 1 and N = 8: weak scaling compares identical per-GPU work
    (at N = 1 the schedule makes no measurable difference: profiles/r05_ab_gather_every.txt)."""
    return 2


def gather_window(t, every, start=0):
    """After step t has been submitted (steps of the current phase count from `start`): (first_step, n_steps) of the
    collective that is due now, or None."""
    every = max(1, int(every))
    return (t - every + 1, every) if (t - start + 1) % every == 0 else None


def gather_tail(end, every, start=0):
    """The steps of the phase [start, end) that no window has covered: (first_step, n) or None."""
    rem = (end - start) % max(1, int(every))
    return (end - rem, rem) if rem else None


def max_over_ranks(seconds, world, device="cpu"):
    import torch
    import torch.distributed as dist
    if world == 1 and not _group_live():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, world, device="cpu"):
    """Every rank's float in rank order (on every rank): the per-rank step times of a bench line, so that a straggler
    GPU shows in the first line of a real multi-GPU run instead of hiding inside the maximum."""
    import torch
    import torch.distributed as dist
    if world == 1 and not _group_live():
        return [float(value)]
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    out = torch.empty(dist.get_world_size(), dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, mine)
    return [float(x) for x in out.cpu().tolist()]


def ranks_seen(world, device="cpu"):
    """Every rank's id as the backend delivers it (one all_gather of one int64 per rank) plus the group's own idea of
    its size: a SCALE line carries this as proof that N ranks really took part.  -> (list of ranks, world size)."""
    import torch
    import torch.distributed as dist
    if not _group_live():
        return [0], 1
    mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device=device)
    out = torch.empty(dist.get_world_size(), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine)
    return [int(x) for x in out.cpu().tolist()], int(dist.get_world_size())


def records_to_poses(buf, n):
    """uint8 numpy/bytes of n records -> (R [n,3,3] row-major, T [n,3], err [n])."""
    a = np.frombuffer(bytes(buf), dtype=np.float32).reshape(n, RECORD_BYTES // 4)
    R = a[:, :9].reshape(n, 3, 3).transpose(0, 2, 1).copy()
    return R, a[:, 9:12].copy(), a[:, 12].copy()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rendezvous_env(rank, world, local_rank=None, port=None, base=None):
    """Environment of one rank of a single-node job (torch.distributed env:// rendezvous on 127.0.0.1)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank if local_rank is None else local_rank), WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port or free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only mode the host driver supports
    return env


def spawn_ranks(script, argv, world, timeout=None):
    """`python script argv` once per rank (one process per GPU: LOCAL_RANK = rank), rank 0 inheriting this
    process's stdout.  Returns the exit codes; a rank that fails takes the others down."""
    port = free_port()
    procs = []
    for r in range(world):
        env = rendezvous_env(r, world, port=port)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [None] * world
    try:
        import time
        t0 = time.time()
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    codes[i] = p.poll()
            if any(c not in (None, 0) for c in codes) or (timeout and time.time() - t0 > timeout):
                break
            time.sleep(0.05)
    finally:
        for i, p in enumerate(procs):  # exactly the processes started here, by handle
            if p.poll() is None:
                p.terminate()
                try:
                    p.wait(10)
                except subprocess.TimeoutExpired:
                    p.kill()
            codes[i] = p.returncode
    return codes
