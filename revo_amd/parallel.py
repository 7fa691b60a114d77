"""Multi-GPU layer of the batched mode: one process per GPU, frame-pairs
sharded statically across ranks, NO data-path collective; the only exchange is one
all_gather of the fixed 96-byte pair records (RCCL over xGMI on GPUs, gloo in the CPU
tests).  At 96 B x pairs the collective is latency-bound; nothing is reduced."""
import numpy as np

RECORD_BYTES = 96


def shard_pairs(global_pairs, rank, world):
    """Static block partition: rank g owns pairs [g*B/G, (g+1)*B/G)."""
    if global_pairs % world:
        raise ValueError("global_pairs must be divisible by the world size")
    per = global_pairs // world
    return list(range(rank * per, (rank + 1) * per))


def gather_records(local_records, world, group=None, out=None):
    """local_records: uint8 tensor [pairs*96] on this rank's device -> uint8 tensor
    [world*pairs*96] holding every rank's records in rank order (on every rank).
    `out`: optional preallocated result tensor (steady-state loops reuse one)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_records
    if out is None:
        out = torch.empty(world * local_records.numel(), dtype=torch.uint8, device=local_records.device)
    elif out.numel() != world * local_records.numel() or out.dtype != torch.uint8:
        raise ValueError("out must be a uint8 tensor of world * len(local_records) elements")
    dist.all_gather_into_tensor(out, local_records, group=group)
    return out


def max_over_ranks(seconds, world, device="cpu"):
    import torch
    import torch.distributed as dist
    if world == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def records_to_poses(buf, n):
    """uint8 numpy/bytes of n records -> (R [n,3,3] row-major, T [n,3], err [n])."""
    a = np.frombuffer(bytes(buf), dtype=np.float32).reshape(n, RECORD_BYTES // 4)
    R = a[:, :9].reshape(n, 3, 3).transpose(0, 2, 1).copy()
    return R, a[:, 9:12].copy(), a[:, 12].copy()
