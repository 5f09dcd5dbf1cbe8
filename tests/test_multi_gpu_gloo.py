"""The N>1 path on CPU: two processes (gloo), each runs the device code (SIMT emulator) on its
contiguous shard, then ONE all-reduce merges the counter blocks - the result must equal a
single engine that saw the whole input.  With the plain per-shard submit the duplicate counters are
the exception (cross-shard copies are not seen); multigpu.run_shard's two-pass protocol removes it."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import engines
    import synth
    from fastp_amd import abi, multigpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = abi.default_params(True, 100)
    p.cut_right = 1
    d = synth.synth_pairs(900, L=100, seed=3, insert_mean=130.0, insert_sd=40.0)
    lo, hi = multigpu.shard_bounds(900, world, rank)
    eng = engines.sim_engine(p)
    sl = {k: v[lo:hi] for k, v in d.items()}
    res = eng.process(sl["seq1"], sl["qual1"], sl["len1"], sl["seq2"], sl["qual2"], sl["len2"])
    merged = multigpu.allreduce_counters_host(eng.counters(), dist)
    eng.close()
    ret[rank] = (merged, res[0].tobytes(), res[1].tobytes(), res[2].tobytes())
    dist.destroy_process_group()


def test_two_rank_shards_merge_to_single_engine_counters():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import engines
    import synth
    from fastp_amd import abi, multigpu
    engines.build_sim()
    assert multigpu.shard_bounds(10, 3, 0) == (0, 3) and multigpu.shard_bounds(10, 3, 2) == (6, 10)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    p = abi.default_params(True, 100)
    p.cut_right = 1
    d = synth.synth_pairs(900, L=100, seed=3, insert_mean=130.0, insert_sd=40.0)
    eng = engines.sim_engine(p)
    whole = eng.process(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    ctr = eng.counters()
    lay = eng.layout
    eng.close()
    m0, m1 = ret[0][0], ret[1][0]
    assert np.array_equal(m0, m1), "all ranks must hold the same merged block"
    keep = np.ones(lay.total, dtype=bool)
    keep[lay.dup_count] = False      # cross-shard duplicates are not seen (documented)
    assert np.array_equal(m0[keep], ctr[keep])
    assert m0[lay.dup_count] <= ctr[lay.dup_count]
    # per-read results are per-shard and concatenate in shard order (dup flag excepted)
    for k in range(3):
        cat = ret[0][k + 1] + ret[1][k + 1]
        a = np.frombuffer(cat, dtype=whole[k].dtype).copy()
        b = whole[k].copy()
        if k < 2:
            a["flags"] &= ~np.uint8(abi.RF_DUP)
            b["flags"] &= ~np.uint8(abi.RF_DUP)
        assert a.tobytes() == b.tobytes()


import pytest


@pytest.mark.parametrize("name,npacks", [("pe_noadapter_dedup", 3), ("pe_overrep", 2), ("se_overrep", 1), ("pe_exotic_dedup_adapters", 2)])
def test_two_rank_exact_protocol_equals_one_stream(name, npacks):
    """run_shard (dup scan -> bitmap all-gather -> prefix -> worker loop -> deferred overrepresentation):
    every record and every counter - duplicates and sampled positions included - equals ONE stream"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import engines
    import oraclelib
    import shard_util
    engines.build_sim()
    n = 1100
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + 7
    if name == "pe_overrep":   # the other shape of the bitmap exchange
        os.environ["FASTP_SHARD_EXCHANGE"] = "allgather"
    try:
        mp.spawn(shard_util.shard_worker, args=(2, port, ret, "sim", name, n, npacks), nprocs=2, join=True)
    finally:
        os.environ.pop("FASTP_SHARD_EXCHANGE", None)
    params, d, paired = shard_util.case_input(name, n)
    o = oraclelib.Oracle(params)
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    whole = o.process(*args)
    ctr = o.counters()
    lay = o.layout
    o.close()
    assert np.array_equal(ret[0][0], ret[1][0])
    assert ctr[lay.dup_count] > 0
    bad = np.nonzero(ret[0][0] != ctr)[0]
    assert len(bad) == 0, f"{name}: counters differ at {bad[:8]}"
    for k in range(3 if paired else 1):
        assert ret[0][k + 1] + ret[1][k + 1] == whole[k].tobytes(), f"{name}: records {k} differ"


def test_two_rank_exact_protocol_after_reset_is_a_fresh_stream():
    """two sharded runs on the same engines with GpuEngine.reset() in between (what bench.py does between cycles): the
    second run's overrepresentation sampling must start at the origin again, i.e. its counters equal ONE fresh stream"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import engines
    import oraclelib
    import shard_util
    engines.build_sim()
    n = 1100
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + 13
    mp.spawn(shard_util.shard_worker, args=(2, port, ret, "sim", "pe_overrep", n, 2, 100, True, 2), nprocs=2, join=True)
    params, d, paired = shard_util.case_input("pe_overrep", n)
    o = oraclelib.Oracle(params)
    whole = o.process(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    ctr = o.counters()
    o.close()
    assert np.array_equal(ret[0][0], ret[1][0])
    bad = np.nonzero(ret[0][0] != ctr)[0]
    assert len(bad) == 0, f"second run after reset: counters differ at {bad[:8]}"
    for k in range(3):
        assert ret[0][k + 1] + ret[1][k + 1] == whole[k].tobytes()


def test_two_rank_plain_submit_misses_cross_shard_duplicates():
    """the input of the exact-protocol test does contain cross-shard duplicates: without the protocol they are missed"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import engines
    import oraclelib
    import shard_util
    engines.build_sim()
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + 11
    mp.spawn(shard_util.shard_worker, args=(2, port, ret, "sim", "pe_default", 1100, 2, 100, False), nprocs=2, join=True)
    params, d, paired = shard_util.case_input("pe_default", 1100)
    o = oraclelib.Oracle(params)
    o.process(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    ctr, lay = o.counters(), o.layout
    o.close()
    assert ret[0][0][lay.dup_count] < ctr[lay.dup_count]


@pytest.mark.parametrize("seed", [209])
def test_two_rank_exact_protocol_on_random_option_sets(seed):
    """the protocol on random option sets (merge, merge + overrepresentation + correction, --dedup + overrepresentation,
    single end): still ONE stream"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import engines
    import oraclelib
    import shard_util
    engines.build_sim()
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + 17 + seed % 50
    mp.spawn(shard_util.shard_worker, args=(2, port, ret, "sim", f"fuzz:{seed}", 0, 2), nprocs=2, join=True)
    params, d, paired = shard_util.case_input(f"fuzz:{seed}", 0)
    o = oraclelib.Oracle(params)
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    whole = o.process(*args)
    ctr = o.counters()
    o.close()
    assert np.array_equal(ret[0][0], ctr), f"seed {seed}: counters differ at {np.nonzero(ret[0][0] != ctr)[0][:8]}"
    for k in range(3 if paired else 1):
        assert ret[0][k + 1] + ret[1][k + 1] == whole[k].tobytes(), f"seed {seed}: records {k} differ"
