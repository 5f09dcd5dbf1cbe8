"""Multi-GPU host logic: one process per GPU, read batches sharded with no data-path
collective, and ONE all-reduce (RCCL over xGMI; `nccl` backend of torch.distributed) of the
int64 counter block at the end - the device analogue of Stats::merge / FilterResult::merge
(stats.cpp:877-955, filterresult.cpp:38-89; call sites peprocessor.cpp:217-234).

Duplicate detection is per shard ("replicas only" for that one feature): every rank keeps
its own bloom bitmaps, so duplicates whose copies land on different GPUs are not seen; the
summed dup_total / dup_count give a lower bound of the single-stream rate.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """contiguous shard [lo, hi) of rank `rank` (GPU g gets pairs [g*N/G, (g+1)*N/G))"""
    return (n * rank) // world, (n * (rank + 1)) // world


def allreduce_counters_host(counters: np.ndarray, dist) -> np.ndarray:
    """CPU / gloo variant (tests): sum the counter blocks of all ranks, keep the header words."""
    import torch
    t = torch.from_numpy(counters.copy())
    hdr = t[:4].clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t[:4] = hdr
    return t.numpy()


def allreduce_counters_device(engine, dist, device) -> None:
    """GPU / RCCL variant: export the engine's counter block into a torch tensor, all-reduce it
    (int64 sum), import the merged block back.  ~0.25 MB: latency bound, not link bound."""
    import torch
    buf = torch.empty(engine.layout.total, dtype=torch.int64, device=device)
    engine.counters_export(buf.data_ptr())
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize(device)
    engine.counters_import(buf.data_ptr())
