"""Multi-GPU: independent samples are sharded over ranks (one process per GPU); no collective on the
data path, ONE gather at the end (RCCL over xGMI when the backend is "nccl").  SURVEY.md section 8e.

The noise stream is keyed by the GLOBAL sample index, every shard pads to the batch-wide N, and every shard
plans its kernels with the WHOLE batch's graph figures (Engine.plan_hint_for / gaudi_set_plan_hint: the
kernel family and the edge-GEMM arithmetic follow from batch-wide maxima), so the gathered result is
identical to the unsharded run.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(total: int, rank: int, world: int):
    """Contiguous block of global sample indices [lo, hi) owned by ``rank``."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sample_sharded(sample_fn, node_mask: np.ndarray, edge_mask: np.ndarray, rank: int, world: int, engine=None):
    """Run ``sample_fn(node_mask_shard, edge_mask_shard, sample_offset) -> (x, h)`` on this rank's block.

    node_mask [B,N(,1)], edge_mask reshapeable to [B,N,N] describe the WHOLE logical batch (already padded
    to the batch-wide N, as sampling_edm.sample_guidance does, sampling_edm.py:177).  ``engine``: the
    gaudi_amd.engine.Engine that sample_fn runs on -- it is told the whole batch's plan figures for the
    duration of the call, so a heterogeneous batch gives the same bits however it is cut."""
    nm = np.asarray(node_mask, np.float32)
    B, N = nm.shape[0], nm.shape[1]
    nm = nm.reshape(B, N)
    em = np.asarray(edge_mask, np.float32).reshape(B, N, N)
    lo, hi = shard_bounds(B, rank, world)
    if hi == lo:
        return lo, hi, None, None
    prev = None
    if engine is not None:
        prev = getattr(engine, "_plan_hint", (0, 0))  # a hint the caller had set is restored afterwards
        engine.set_plan_hint(*engine.plan_hint_for(nm, em))
    try:
        x, h = sample_fn(nm[lo:hi], em[lo:hi], lo)
    finally:
        if engine is not None:
            engine.set_plan_hint(*prev)
    return lo, hi, x, h


def check_same_plan(engine, device=None):
    """All ranks must have run the same kernel family / edge-GEMM arithmetic (one small all_gather of two ints).
    Raises if they differ: the gathered batch would then mix roundings (pass ``engine=`` to sample_sharded)."""
    import torch
    import torch.distributed as dist

    mine = torch.tensor([engine.kernel_variant()[1], min(engine.edge_math()[1], 1)], dtype=torch.int32)
    if device is not None:
        mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    plans = sorted({tuple(int(v) for v in t.cpu()) for t in out})
    if len(plans) != 1:
        raise RuntimeError(f"ranks ran different kernel plans (waves, split edge GEMMs): {plans}")
    return plans[0]


def gather_to_all(x_local, h_local, total: int, N: int, F: int, device=None):
    """ONE collective over the per-rank results (padded to the largest shard) -> full [total,N,3], [total,N,F].

    With ``device`` (the RCCL path): the shard is written once into a pinned host tensor, goes to the device in one copy, ONE
    all_gather_into_tensor fills a single [world, per, N, 3+F] device tensor, and one copy brings it back (the C ABI hands results
    over in host memory, include/gaudi_hip.h: gaudi_sample).  Without it (gloo, CPU tests): a list all_gather of host tensors."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    per = max(shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world))
    D = 3 + F
    if device is not None:
        send = torch.zeros((per, N, D), dtype=torch.float32, pin_memory=True)
        if x_local is not None:
            n = x_local.shape[0]
            send[:n, :, :3] = torch.from_numpy(np.ascontiguousarray(x_local, np.float32))
            send[:n, :, 3:] = torch.from_numpy(np.ascontiguousarray(h_local, np.float32))
        dev_send = send.to(device, non_blocking=True)
        dev_recv = torch.empty((world, per, N, D), dtype=torch.float32, device=device)
        dist.all_gather_into_tensor(dev_recv, dev_send)
        full = dev_recv.cpu().numpy()
    else:
        buf = np.zeros((per, N, D), np.float32)
        if x_local is not None:
            n = x_local.shape[0]
            buf[:n, :, :3] = x_local
            buf[:n, :, 3:] = h_local
        t = torch.from_numpy(buf)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        full = np.stack([o.numpy() for o in out], 0)
    xs, hs = [], []
    for r in range(world):
        lo, hi = shard_bounds(total, r, world)
        a = full[r, : hi - lo]
        xs.append(a[:, :, :3])
        hs.append(a[:, :, 3:])
    return np.concatenate(xs, 0), np.concatenate(hs, 0)
