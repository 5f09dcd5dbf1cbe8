"""The C ABI's own collectives (fastp_amd/csrc/fq_comm.cpp: fastp_gpu_comm_init, fastp_gpu_allreduce,
fastp_gpu_exchange_dup_prefix) with n = 2 contexts in one process: the product source compiled into the emulator
library, librccl replaced by the in-process stand-in tests/rccl_stub (FASTP_GPU_RCCL_LIB).  What this exercises
without a second GPU: argument marshalling, the header words of the counter block surviving the sum, the
send / recv schedule of the bitmap exchange (transpose - scan - transpose), error paths.  The result must be the
one-stream answer of the oracle, exactly as for the torch.distributed path of test_multi_gpu_gloo.py."""
import ctypes as C
import os

import numpy as np
import pytest

import engines
import oraclelib
import shard_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "rccl_stub", "librccl_stub.so")


def _pair_of_engines(params):
    engines.build_sim()
    assert os.path.exists(STUB)
    os.environ["FASTP_GPU_RCCL_LIB"] = STUB
    e0, e1 = engines.sim_engine(params), engines.sim_engine(params)
    cid = e0.comm_id()
    e0.comm_init(cid, 2, 0)
    e1.comm_init(cid, 2, 1)
    return e0, e1


def _both(e0, e1, name):
    lib = e0.lib
    fn = getattr(lib, name)
    fn.argtypes = [C.c_void_p, C.c_int]
    arr = (C.c_void_p * 2)(e0.h, e1.h)
    rc = fn(arr, 2)
    if rc != 0:
        lib.fastp_gpu_comm_last_error.restype = C.c_char_p
        raise AssertionError(f"{name} -> {rc}: {(lib.fastp_gpu_comm_last_error() or b'').decode()}")


@pytest.mark.parametrize("name", ["pe_default"])   # (accuracy level 1: the emulator ORs 1 GiB images; --dedup is the gloo tests')
def test_cabi_collectives_two_contexts_equal_one_stream(name):
    import torch
    from fastp_amd import multigpu
    n = 1100
    params, d, paired = shard_util.case_input(name, n)
    e0, e1 = _pair_of_engines(params)
    dev = torch.device("cpu")
    keeps = []
    scans = []
    for rank, eng in enumerate((e0, e1)):
        lo, hi = multigpu.shard_bounds(n, 2, rank)
        batches, results, keep = shard_util.device_batches(eng, d, lo, hi, 2, dev)
        sc = [torch.zeros(max(16, eng.dup_scan_bytes(b.n)), dtype=torch.uint8) for b in batches]
        for b, r, t in zip(batches, results, sc):
            eng.submit_pass1_device(b, t.data_ptr(), r)
        eng.synchronize()
        keeps.append((batches, results, keep))
        scans.append(sc)
    _both(e0, e1, "fastp_gpu_exchange_dup_prefix")          # rank r <- OR of the bitmaps of ranks < r
    for (batches, results, keep), sc, eng in zip(keeps, scans, (e0, e1)):
        for b, r, t in zip(batches, results, sc):
            eng.submit_pass2_device(b, t.data_ptr(), r)
        eng.synchronize()
    before = [e.counters() for e in (e0, e1)]
    _both(e0, e1, "fastp_gpu_allreduce")                    # Stats::merge / FilterResult::merge
    merged = [e.counters() for e in (e0, e1)]
    o = oraclelib.Oracle(params)
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    whole = o.process(*args)
    ctr = o.counters()
    lay = o.layout
    o.close()
    assert np.array_equal(merged[0], merged[1]), "every rank holds the same merged block"
    assert np.array_equal(merged[0][:4], before[0][:4]), "header words (ABI version, cycles, insert-size bound) are kept, not summed"
    assert ctr[lay.dup_count] > 0
    bad = np.nonzero(merged[0] != ctr)[0]
    assert len(bad) == 0, f"{name}: merged counters differ from one stream at {bad[:8]}"
    recs = [shard_util.fetch_records(k[2], paired) for k in keeps]
    for k in range(3 if paired else 1):
        assert recs[0][k] + recs[1][k] == whole[k].tobytes(), f"{name}: records {k} differ"
    e0.close()
    e1.close()


def test_cabi_collectives_error_paths():
    from fastp_amd import abi
    p = abi.default_params(True, 100)
    p.dup_enabled = 0
    engines.build_sim()
    os.environ["FASTP_GPU_RCCL_LIB"] = STUB
    os.environ["FASTP_STUB_TIMEOUT_MS"] = "300"   # how long the stand-in waits for a rank that never shows up
    e = engines.sim_engine(p)
    with pytest.raises(Exception):
        e.allreduce()                       # no communicator yet
    cid = e.comm_id()
    e.comm_init(cid, 2, 0)                  # a 2-rank communicator of which only one rank shows up
    with pytest.raises(Exception):
        e.allreduce()                       # the stand-in reports the missing rank; the group is closed again
    e.comm_init(e.comm_id(), 1, 0)          # re-initialising drops the old communicator
    e.allreduce()
    c = e.counters()
    assert c[0] == abi.ABI_VERSION if hasattr(abi, "ABI_VERSION") else c[0] > 0
    e.close()
    os.environ.pop("FASTP_STUB_TIMEOUT_MS", None)


def test_comm_destroy_waits_for_a_collective_in_flight():
    """fastp_gpu_comm_destroy from another thread while fastp_gpu_allreduce is inside the (stand-in) RCCL call: the registry
    entry is pinned for the length of the call, so the destroy returns only after the collective has (here: with the
    stand-in's 'rank missing' error) - it used to erase the entry under the call."""
    import threading
    import time
    from fastp_amd import abi
    p = abi.default_params(True, 100)
    p.dup_enabled = 0
    engines.build_sim()
    os.environ["FASTP_GPU_RCCL_LIB"] = STUB
    os.environ["FASTP_STUB_TIMEOUT_MS"] = "1500"
    e = engines.sim_engine(p)
    e.comm_init(e.comm_id(), 2, 0)          # the other rank never shows up: the collective sits in the stand-in for 1.5 s
    done = {}

    def collective():
        try:
            e.allreduce()
            done["rc"] = "ok"
        except Exception as ex:             # noqa: BLE001
            done["rc"] = str(ex)
        done["t"] = time.time()

    th = threading.Thread(target=collective)
    t0 = time.time()
    th.start()
    time.sleep(0.4)                          # the collective is inside the library by now
    e.lib.fastp_gpu_comm_destroy.argtypes = [C.c_void_p]
    e.lib.fastp_gpu_comm_destroy.restype = None
    e.lib.fastp_gpu_comm_destroy(e.h)
    t_destroy = time.time()
    th.join()
    assert done["rc"] != "ok"
    assert t_destroy >= done["t"] - 0.05 and t_destroy - t0 > 1.0, (t_destroy - t0, done["t"] - t0)
    with pytest.raises(Exception):
        e.allreduce()                        # the communicator is gone
    e.close()
    os.environ.pop("FASTP_STUB_TIMEOUT_MS", None)


def test_run_shard_with_the_cabi_exchange_two_ranks_as_threads():
    """multigpu.run_shard(exchange="cabi") - what bench.py --gpus N does by default: pass 1, fastp_gpu_exchange_dup_prefix,
    pass 2 - with the two ranks as two threads of this process (one context each, the stand-in librccl makes them meet);
    then fastp_gpu_allreduce.  Records and merged counters are the oracle's one stream."""
    import threading
    import torch
    from fastp_amd import multigpu
    n = 1300
    params, d, paired = shard_util.case_input("pe_default", n)
    os.environ["FASTP_STUB_TIMEOUT_MS"] = "240000"   # the emulator runs one launch at a time: a rank may wait for the other's kernels
    e0, e1 = _pair_of_engines(params)
    dev = torch.device("cpu")
    keeps, errs, timings = [None, None], [], [{}, {}]

    def rank_main(rank, eng):
        try:
            lo, hi = multigpu.shard_bounds(n, 2, rank)
            batches, results, keep = shard_util.device_batches(eng, d, lo, hi, 2, dev)
            keeps[rank] = (batches, results, keep)
            multigpu.run_shard(eng, None, rank, 2, batches, results, dev, force=True, exchange="cabi", timings=timings[rank])
            eng.allreduce()
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=rank_main, args=(r, e)) for r, e in enumerate((e0, e1))]
    for t in ths:
        t.start()
    for t in ths:
        t.join(300)
    assert not errs, errs
    assert all("exchange_s" in t for t in timings)
    merged = [e.counters() for e in (e0, e1)]
    o = oraclelib.Oracle(params)
    whole = o.process(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    ctr = o.counters()
    lay = o.layout
    o.close()
    assert ctr[lay.dup_count] > 0
    assert np.array_equal(merged[0], merged[1])
    bad = np.nonzero(merged[0] != ctr)[0]
    assert len(bad) == 0, f"merged counters differ from one stream at {bad[:8]}"
    recs = [shard_util.fetch_records(k[2], True) for k in keeps]
    for k in range(3):
        assert recs[0][k] + recs[1][k] == whole[k].tobytes(), f"records {k} differ"
    e0.close()
    e1.close()
