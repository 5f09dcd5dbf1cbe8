"""helpers for the sharded-run tests (fastp_amd.multigpu.run_shard): packed batches / result buffers as torch
tensors on a device (cpu for the SIMT emulator, cuda for the real engine)"""
import ctypes as C

import numpy as np

from fastp_amd import abi, engine


def device_batches(eng, d, lo, hi, npacks, device, corr_cap=1 << 16):
    """shard [lo, hi) of the ASCII arrays `d`, cut into `npacks` batches resident on `device`"""
    import torch
    paired = "seq2" in d
    ml = eng.params.max_len
    batches, results, keep = [], [], []
    edges = [lo + (hi - lo) * k // npacks for k in range(npacks + 1)]
    for a, e in zip(edges[:-1], edges[1:]):
        n = e - a
        b = abi.Batch()
        b.n, b.flags = n, abi.BATCH_STAT_ISIZE
        tens = {}
        mask = np.zeros(n, dtype=np.uint8)   # units with letters outside ACGTN: listed in the batch with their raw rows
        for m in ("1", "2") if paired else ("1",):
            s, q, l = engine.pack_ascii(eng.lib, ml, d["seq" + m][a:e], d["qual" + m][a:e], d["len" + m][a:e], mask)
            for nm, arr in (("seq", s), ("qual", q), ("len", l)):
                t = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()).to(device)
                if t.numel() == 0:
                    t = torch.zeros(16, dtype=torch.uint8, device=device)
                tens[nm + m] = t
        b.seq1, b.qual1, b.len1 = tens["seq1"].data_ptr(), tens["qual1"].data_ptr(), tens["len1"].data_ptr()
        if paired:
            b.seq2, b.qual2, b.len2 = tens["seq2"].data_ptr(), tens["qual2"].data_ptr(), tens["len2"].data_ptr()
        if mask.any():
            xu = np.flatnonzero(mask).astype(np.int32)
            tens["xunit"] = xu   # host memory, kept alive with the batch
            b.n_exotic, b.exotic_unit = len(xu), xu.ctypes.data
            for k, m in enumerate(("1", "2") if paired else ("1",)):
                rows = np.ascontiguousarray(np.asarray(d["seq" + m][a:e], dtype=np.uint8)[xu])
                tens["xtext" + m] = torch.from_numpy(rows.reshape(-1).copy()).to(device)
                tens["xoff" + m] = torch.from_numpy((np.arange(len(xu), dtype=np.uint32) * np.uint32(rows.shape[1])).view(np.uint8).copy()).to(device)
                b.exotic_text[k], b.exotic_off[k] = tens["xtext" + m].data_ptr(), tens["xoff" + m].data_ptr()
        r = abi.Results()
        out = dict(r1=torch.zeros(max(1, n) * 12, dtype=torch.uint8, device=device),
                   r2=torch.zeros(max(1, n) * 12, dtype=torch.uint8, device=device),
                   pair=torch.zeros(max(1, n) * 8, dtype=torch.uint8, device=device),
                   corr=torch.zeros(corr_cap * 8, dtype=torch.uint8, device=device),
                   nc=torch.zeros(4, dtype=torch.int32, device=device),
                   ev=torch.zeros((4 * n + 16) * 12, dtype=torch.uint8, device=device),
                   nev=torch.zeros(4, dtype=torch.int32, device=device))
        r.r1 = out["r1"].data_ptr()
        if paired:
            r.r2, r.pair = out["r2"].data_ptr(), out["pair"].data_ptr()
        r.corrections, r.corrections_capacity, r.n_corrections = out["corr"].data_ptr(), corr_cap, out["nc"].data_ptr()
        r.adapter_events, r.adapter_events_capacity, r.n_adapter_events = out["ev"].data_ptr(), 4 * n + 16, out["nev"].data_ptr()
        batches.append(b)
        results.append(r)
        keep.append((tens, out, n))
    return batches, results, keep


def fetch_records(keep, paired):
    """concatenated (r1, r2, pair) bytes of all batches"""
    r1 = b"".join(o["r1"][:n * 12].cpu().numpy().tobytes() for _, o, n in keep)
    r2 = b"".join(o["r2"][:n * 12].cpu().numpy().tobytes() for _, o, n in keep) if paired else b""
    pr = b"".join(o["pair"][:n * 8].cpu().numpy().tobytes() for _, o, n in keep) if paired else b""
    return r1, r2, pr


def shard_worker(rank, world, port, ret, kind, name, n, npacks, L=100, exact=True, runs=1):
    """one rank of a sharded run over gloo; kind = 'sim' (emulator, cpu tensors) or 'gpu' (cuda:0 for every rank)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch
    import torch.distributed as dist
    import engines
    from fastp_amd import multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, d, paired = case_input(name, n, L)
    n = len(d["len1"])
    device = torch.device("cpu") if kind == "sim" else torch.device("cuda", 0)
    eng = engines.sim_engine(params) if kind == "sim" else engines.gpu_engine(params)
    lo, hi = multigpu.shard_bounds(n, world, rank)
    batches, results, keep = device_batches(eng, d, lo, hi, npacks, device)
    for k in range(runs):   # runs > 1: the same shard again after GpuEngine.reset() - every run must be a fresh stream
        if k:
            eng.reset()
        multigpu.run_shard(eng, dist, rank, world, batches, results, device, exact=exact)
    merged = multigpu.allreduce_counters_host(eng.counters(), dist)
    recs = fetch_records(keep, paired)
    eng.close()
    ret[rank] = (merged,) + recs
    dist.destroy_process_group()


def case_input(name, n, L=100):
    import cases
    import synth
    if name.startswith("fuzz:"):   # a random option set of tests/test_option_fuzz.py (n and L come with it)
        import test_option_fuzz
        p, d, paired = test_option_fuzz.random_case(int(name[5:]))
        p.dup_accuracy_level = 1
        return p, d, paired
    paired, flags, pf, skw = cases.CASES[name]
    skw = dict(skw)
    skw.setdefault("dup_frac", 0.3)
    d = synth.synth_pairs(n, L=L, seed=19, paired=paired, **skw)
    p = pf(L)
    p.dup_accuracy_level = 1   # 2 x 512 MiB bitmaps: the exchange buffers of two ranks must fit the test box
    params = cases.finalize_params(name, p, d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    return params, d, paired


def device_bit_positions(eng, d, device):
    """Duplicate::seq2intvector mod mBufLenInBits of every unit as the DEVICE computed it: the scan state the
    sharded protocol's pass 1 leaves behind ([n][bufnum] u64 first)"""
    import torch
    n = len(d["len1"])
    batches, results, keep = device_batches(eng, d, 0, n, 1, device)
    scan = torch.zeros(max(16, eng.dup_scan_bytes(n)), dtype=torch.uint8, device=device)
    eng.submit_pass1_device(batches[0], scan.data_ptr(), results[0])
    eng.synchronize()
    B = {1: 2, 2: 2, 3: 4, 4: 4, 5: 4, 6: 8}[int(eng.params.dup_accuracy_level)]
    return scan[:n * B * 8].cpu().numpy().view(np.uint64).reshape(n, B).copy()


def oracle_bit_positions(level, d):
    import oraclelib
    n = len(d["len1"])
    B = {1: 2, 2: 2, 3: 4, 4: 4, 5: 4, 6: 8}[int(level)]
    pos = np.zeros((n, B), dtype=np.uint64)
    paired = "seq2" in d
    s1 = np.ascontiguousarray(d["seq1"], dtype=np.uint8)
    l1 = np.ascontiguousarray(d["len1"], dtype=np.int32)
    s2 = np.ascontiguousarray(d["seq2"], dtype=np.uint8) if paired else None
    l2 = np.ascontiguousarray(d["len2"], dtype=np.int32) if paired else None
    oraclelib.lib().fastp_oracle_dup_bits_batch(int(level), n, int(s1.shape[1]), s1.ctypes.data, l1.ctypes.data,
                                                s2.ctypes.data if paired else None, l2.ctypes.data if paired else None,
                                                pos.ctypes.data)
    return pos
