"""Multi-GPU host logic: one process per GPU, read batches sharded with no data-path
collective, and ONE all-reduce (RCCL over xGMI; `nccl` backend of torch.distributed) of the
int64 counter block at the end - the device analogue of Stats::merge / FilterResult::merge
(stats.cpp:877-955, filterresult.cpp:38-89; call sites peprocessor.cpp:217-234).

Two quantities of the worker loop depend on what came EARLIER in the stream - the duplicate
bloom filter (duplicate.cpp:122-163) and the overrepresentation sampling positions
(stats.cpp:272) - and `run_shard` reproduces both exactly for a sharded run: a
duplicate scan pass, one all-gather of the bloom bitmaps (the only bandwidth-bound exchange:
mBufNum * mBufLenInBytes per rank, 1 GiB at the default accuracy level), the worker loop with
the decision taken against the OR of the preceding shards' bitmaps, and a tiny all-gather of
stream positions before the deferred overrepresentation pass.  The merged counters and every
result record are then those of ONE stream over the concatenated shards (`-w 1` semantics).
Without that protocol (plain `submit_device` per rank) duplicates whose copies land on different
GPUs are not seen and the summed dup counters are only a lower bound.
"""
from __future__ import annotations

import os

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
    _device_sync(buf)
    engine.counters_import(buf.data_ptr())


def _all_gather_flat(dist, mine, world):
    """all_gather_into_tensor; a gloo group (CPU tests, single-GPU rehearsals) gets host tensors"""
    import torch
    staged = mine.is_cuda and dist.get_backend() != "nccl"
    src = mine.cpu() if staged else mine
    out = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)
    dist.all_gather_into_tensor(out, src)
    _device_sync(out)
    return out.to(mine.device) if staged else out


def _device_sync(t):
    """RCCL work is enqueued on the process group's own stream, which torch's current stream is made to wait for, and the
    engine launches on ITS own stream: a collective's output must be complete before an engine call reads it.  Waiting
    for torch's current stream (an event behind the collective) is enough - no device-wide synchronisation."""
    if t.is_cuda:
        import torch
        torch.cuda.current_stream(t.device).synchronize()


def run_shard(engine, dist, rank: int, world: int, batches, results, device, *, exact=True, force=False, scans=None, exchange="torch",
              timings=None):
    """Process this rank's shard (`batches[i]` -> `results[i]`, abi.Batch / abi.Results holding
    pointers valid on `device`: HBM for the real engine) so that results and counters equal those
    of one stream over all shards in rank order.  `dist` is torch.distributed (nccl = RCCL on the
    GPUs; gloo in the CPU tests) or None for a single shard.  The counter all-reduce is separate
    (allreduce_counters_*).  `scans`: optional preallocated uint8 tensors, one per batch, of at least
    engine.dup_scan_bytes(batch.n) bytes each (otherwise allocated here).  Returns their total size in bytes.
    `exchange`: "cabi" = the bitmap prefix exchange is the engine's own collective (fastp_gpu_exchange_dup_prefix: RCCL
    send / recv groups behind the C ABI; the engine must have a communicator, GpuEngine.comm_init), "torch" = the same
    sequence through torch.distributed (RCCL as well on the GPUs; the only choice for gloo).  `timings`: optional dict,
    gets "exchange_s" added (wall time of the exchange step, synchronised) - a per-run constant, reported apart."""
    import torch
    from . import abi
    if not exact or ((world == 1 or dist is None) and not force):
        for b, r in zip(batches, results):
            engine.submit_device(b, r)
        engine.synchronize()
        return 0
    defer = bool(engine.params.overrep_enabled)
    post_before = 0
    if defer:
        for b in batches:
            b.flags |= abi.BATCH_DEFER_OVERREP
        lay = engine.layout   # the engine's POST1 read counter is cumulative: this call's share is the difference
        post_before = int(engine.counters()[lay.stats[1] + lay.st_reads])
    # pass 1: insert in input order, keep per-unit positions / "set earlier in this shard" masks
    # (without --dedup this already is the whole worker loop, minus the duplicate decision)
    if scans is None:
        scans = [torch.empty(max(16, engine.dup_scan_bytes(b.n)), dtype=torch.uint8, device=device) for b in batches]
    for b, r, t in zip(batches, results, scans):
        engine.submit_pass1_device(b, t.data_ptr(), r)
    engine.synchronize()
    # exchange: exclusive prefix-OR of the bitmaps in rank order
    nbytes = engine.dup_bitmap_bytes()
    import time
    t_x = time.perf_counter()
    if nbytes and world > 1 and exchange == "cabi":
        engine.exchange_dup_prefix()   # slices all-to-all, scan on the slice owner, all-to-all back, prefix set (fq_comm.cpp)
    elif nbytes and world > 1:
        mine = torch.empty(nbytes, dtype=torch.uint8, device=device)
        engine.dup_bitmap_export(mine.data_ptr())
        staged = mine.is_cuda and dist.get_backend() != "nccl"   # gloo rehearsal on a GPU: through host memory
        if nbytes % (16 * world) == 0 and os.environ.get("FASTP_SHARD_EXCHANGE", "alltoall") != "allgather":
            # transpose - scan - transpose: rank s owns slice s of every image, scans it over the ranks and sends
            # each rank its prefix slice back.  2 * (world-1)/world images cross the links per rank, not world-1.
            src = mine.cpu() if staged else mine
            slices = torch.empty_like(src)
            dist.all_to_all_single(slices, src)                 # slices[k] = rank k's image, my slice
            _device_sync(slices)
            if staged:
                slices = slices.to(device)
            engine.prefix_or_images(slices.data_ptr(), world, nbytes // world)
            back = slices.cpu() if staged else slices
            dist.all_to_all_single(src, back)                   # src[s] = my prefix, slice s  -> the whole prefix image
            _device_sync(src)
            prefix = src.to(device) if staged else src
            _device_sync(prefix)
            engine.dup_prefix_set(prefix.data_ptr(), 1)
            del slices, back, prefix, src
        else:
            images = _all_gather_flat(dist, mine, world)
            _device_sync(images)
            engine.dup_prefix_set(images.data_ptr(), rank)
            del images
        del mine
    else:
        engine.dup_prefix_set(None, 0)
    if timings is not None:
        engine.synchronize()
        timings["exchange_s"] = timings.get("exchange_s", 0.0) + time.perf_counter() - t_x
    # pass 2: the decision (and, with --dedup, the worker loop that depends on it)
    for b, r, t in zip(batches, results, scans):
        engine.submit_pass2_device(b, t.data_ptr(), r)
    engine.synchronize()
    if defer:
        lay = engine.layout
        ctr = engine.counters()
        mine = torch.tensor([sum(int(b.n) for b in batches), int(ctr[lay.stats[1] + lay.st_reads]) - post_before], dtype=torch.int64,
                            device=device)
        allpos = _all_gather_flat(dist, mine, world) if world > 1 else mine
        per_rank = allpos.cpu().view(world, 2)
        before = per_rank[:rank].sum(dim=0)
        # a later call on the same engines continues the stream after everything the earlier calls fed to ALL ranks
        done_u, done_p = getattr(engine, "_shard_stream_done", (0, 0))
        engine.stream_set_origin(done_u + int(before[0]), done_p + int(before[1]))
        engine._shard_stream_done = (done_u + int(per_rank[:, 0].sum()), done_p + int(per_rank[:, 1].sum()))
        for b, r in zip(batches, results):
            engine.overrep_device(b, r)
        engine.synchronize()
    return sum(t.numel() for t in scans)
