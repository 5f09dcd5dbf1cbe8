"""Parity tests proper: the HIP path on a real MI355X, called through the C ABI, against
the CPU oracle on the same seeded inputs, against the committed golden vectors of the real
reference, and - at large sizes - through size-independent properties."""
import os

import numpy as np
import pytest

import cases
import driver
import engines
import evalport
import gap_util
import golden_util
import oraclelib
import synth
from fastp_amd import abi, engine

pytestmark = pytest.mark.gpu

SUPPORTED = list(cases.CASES)


def _args(d, paired):
    return (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())


def _compare(name, params, d, paired, n_expect=None):
    o = oraclelib.Oracle(params)
    g = engines.gpu_engine(params)
    ro, rg = o.process(*_args(d, paired)), g.process(*_args(d, paired))
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    for k, what in enumerate(("read1 results", "read2 results", "pair results")):
        if ro[k] is not None:
            bad = np.nonzero(ro[k] != rg[k])[0]
            assert len(bad) == 0, (f"{name}: {what} differ at {len(bad)} of {len(ro[k])} entries, first {bad[:5]}: "
                                   f"oracle {ro[k][bad[:3]]} gpu {rg[k][bad[:3]]}")
    so = np.sort(ro[3], order=["read", "pos"])
    sg = np.sort(rg[3], order=["read", "pos"])
    assert np.array_equal(so, sg), f"{name}: correction lists differ ({len(so)} vs {len(sg)})"
    bad = np.nonzero(co != cg)[0]
    assert len(bad) == 0, f"{name}: {len(bad)} counters differ, first at {bad[:8]}: oracle {co[bad[:8]]} gpu {cg[bad[:8]]}"


@pytest.mark.parametrize("name", SUPPORTED)
def test_gpu_equals_oracle(name):
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(12000, L=150, seed=77, paired=paired, **skw)   # (46 flag sets: the host-side oracle is most of this test's time)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    _compare(name, params, d, paired)


@pytest.mark.parametrize("name", [n for n in golden_util.names()
                                  if n == "testdata_pe" or n in SUPPORTED])
def test_gpu_equals_reference_golden(name):
    """trimmed FASTQ (md5), failed_out and every JSON number the real reference produced"""
    fq1, fq2, meta = golden_util.load(name)
    params = golden_util.params_for(name, max_len=152, fq1=fq1, fq2=fq2)
    eng = engines.gpu_engine(params)
    outs, ctr, rep = driver.run_engine(eng, params, fq1, fq2, umi=golden_util.umi_for(name))
    eng.close()
    golden_util.check_against_golden(name, outs, rep, meta)


@pytest.mark.parametrize("name", ["pe_default", "pe_correction", "pe_merge_unmerged", "pe_allow_gap_indel_corr", "pe_adapter_fasta",
                                  "pe_noadapter_dedup", "pe_overrep", "pe_polyg_polyx", "pe_merge_overlapped_out_trims",
                                  "se_adapter_long_indel", "se_polyx_complexity", "pe_exotic_merge"])
def test_gpu_text_kernel_on_every_unit_equals_reference_golden(name, monkeypatch):
    """FASTP_GPU_EXACT=1: the plan's kernels see only empty reads, every unit goes through the text kernel (fq_text.h) -
    the kernel that takes the units with letters outside ACGTN - and through its compensation of the empty units"""
    monkeypatch.setenv("FASTP_GPU_EXACT", "1")
    fq1, fq2, meta = golden_util.load(name)
    params = golden_util.params_for(name, max_len=152, fq1=fq1, fq2=fq2)
    eng = engines.gpu_engine(params)
    outs, ctr, rep = driver.run_engine(eng, params, fq1, fq2, umi=golden_util.umi_for(name))
    eng.close()
    golden_util.check_against_golden(name, outs, rep, meta)


def test_gpu_sparse_exotic_units_at_scale():
    """80 k pairs, one unit in a thousand with letters outside ACGTN: records and counters equal the oracle's"""
    p = abi.default_params(True, 150)
    p.cut_right = 1
    d = synth.synth_pairs(80000, L=150, seed=91, insert_mean=200.0, exotic_frac=0.0005)
    _compare("sparse_exotic", p, d, True)


@pytest.mark.parametrize("k", range(len(cases.TRIM_STRESS)))
def test_gpu_trim_and_cut_stress(k):
    """Filter::trimAndCut via the predicate-mask bit scans vs the oracle's literal loops, on
    adversarial qualities / N runs / every read length"""
    paired, kw = cases.TRIM_STRESS[k]
    p = abi.default_params(paired, 150)
    if not paired:
        p.adapter_seq_r1 = None
    for key, v in kw.items():
        setattr(p, key, v)
    d = synth.noisy_reads(20000, L=150, seed=100 + k, paired=paired)
    _compare(f"trim_stress{k}", p, d, paired)


@pytest.mark.parametrize("k", range(len(cases.OVERLAP_STRESS)))
def test_gpu_overlap_stress(k):
    """OverlapAnalysis::analyze (prefilter + exact verify + scan-order key) vs the oracle's literal scan:
    every insert size, ragged mates, N on both strands, mismatch bursts after the protected prefix"""
    p = abi.default_params(True, 150)
    for key, v in cases.OVERLAP_STRESS[k].items():
        setattr(p, key, v)
    d = synth.overlap_pairs(20000, L=150, seed=300 + k)
    _compare(f"overlap_stress{k}", p, d, True)


@pytest.mark.parametrize("k", range(len(cases.MERGE_STRESS)))
def test_gpu_merge_stress(k):
    """merge mode: second overlap analysis, OverlapAnalysis::merge, passFilter(merged), stats of the merged read"""
    p = abi.default_params(True, 150)
    for key, v in cases.MERGE_STRESS[k].items():
        setattr(p, key, v)
    d = synth.overlap_pairs(20000, L=150, seed=700 + k, err=0.01, n_rate=0.01)
    _compare(f"merge_stress{k}", p, d, True)


@pytest.mark.twin("test_sim_merge_on_the_lane_plan")
@pytest.mark.parametrize("k", range(len(cases.MERGE_LANE)))
@pytest.mark.parametrize("slow", [0, 1])
def test_gpu_merge_on_the_lane_plan(k, slow, monkeypatch):
    """--merge on the lane plan: the second overlap analysis, the merged read's filter and Stats, --include_unmerged, -c's edits in
    either part; slow: every merged read's second part counted by the lane kernel itself (tests/test_hostsim_parity.py)"""
    if slow:
        monkeypatch.setenv("FASTP_GPU_TEST_MERGE_SLOW", "1")
    p, sets = cases.merge_lane_case(k, n=5000)
    g = engines.gpu_engine(p)
    assert g.plan() == "lane"
    g.close()
    for i, d in enumerate(sets):
        _compare(f"merge_lane{k}/{i}", p, d, True)


def test_gpu_stats_work_list_overflow():
    """one-pass Stats: so many N-containing quality dwords that the LDS work list overflows"""
    p = abi.default_params(True, 150)
    p.cut_right = 1
    p.n_base_limit = 100
    d = synth.noisy_reads(20000, L=150, seed=77, paired=True, n_rate=0.35)
    _compare("stats_overflow", p, d, True)


@pytest.mark.parametrize("name", ["pe_default", "pe_merge", "pe_adapter_fasta", "pe_umi_per_read", "se_adapter_cut",
                                  "pe_noadapter_dedup", "pe_overlapped_out_trims", "pe_merge_overlapped_out",
                                  "pe_merge_overlapped_out_trims"])
def test_gpu_with_cpp_host_glue_equals_reference_golden(name):
    """device records -> C++ host glue (include/fastp_gpu_host.h, fq_glue.cpp) -> the reference's FASTQ + JSON"""
    fq1, fq2, meta = golden_util.load(name)
    params = golden_util.params_for(name, max_len=152, fq1=fq1, fq2=fq2)
    eng = engines.gpu_engine(params)
    outs, ctr, rep = driver.run_engine(eng, params, fq1, fq2, umi=golden_util.umi_for(name), cpp_host_lib=eng.lib)
    eng.close()
    golden_util.check_against_golden(name, outs, rep, meta)


@pytest.mark.parametrize("L", [36, 75, 100, 250, 400])
def test_gpu_read_lengths(L):
    p = abi.default_params(True, L)
    p.cut_right = 1
    p.cut_front = 1
    p.correction = 1
    p.poly_g = 1
    p.poly_x = 1
    d = synth.synth_pairs(6000, L=L, seed=L, insert_mean=L * 1.4, insert_sd=L * 0.5, insert_min=10,
                          insert_max=max(800, 3 * L), polyg_frac=0.1, polyx_frac=0.1)
    _compare(f"L{L}", p, d, True)


def test_gpu_config5_shape_2x250_dedup_overrep():
    """BASELINE configs[4]'s option set on one GPU: PE 2x250, --dedup (accuracy level 3: four bloom buffers) and the
    overrepresentation analysis, against the oracle"""
    from fastp_amd import hostloop
    L = 250
    p = abi.default_params(True, L)
    p.cut_right = 1
    p.dedup = 1
    p.dup_accuracy_level = 3
    d = synth.synth_pairs(12000, L=L, seed=77, insert_mean=260.0, insert_sd=90.0, insert_min=30, insert_max=900,
                          dup_frac=0.25, polyx_frac=0.2)
    b1, b2 = cases._ArrayBatch(d["seq1"], d["len1"]), cases._ArrayBatch(d["seq2"], d["len2"])
    e1, e2 = evalport.evaluate_seq_len(b1), evalport.evaluate_seq_len(b2)
    abi.set_overrep(p, evalport.evaluate_overrep_seqs(b1, e1), evalport.evaluate_overrep_seqs(b2, e2), e1, e2, 5)
    _compare("config5", p, d, True)


def test_gpu_edge_batches():
    p = abi.default_params(True, 150)
    p.cut_tail = 1
    g = engines.gpu_engine(p)
    o = oraclelib.Oracle(p)
    stride = 152
    # empty batch
    e = np.zeros((0, stride), dtype=np.uint8)
    r = g.process(e, e, np.zeros(0, dtype=np.int32), e, e, np.zeros(0, dtype=np.int32))
    assert len(r[0]) == 0
    # single pair, all-empty reads, all-N reads, mixed
    rng = np.random.default_rng(1)
    for kind in ("single", "empty", "allN", "len1", "mixed"):
        n = 1 if kind == "single" else 257
        d = synth.synth_pairs(n, L=150, seed=9, ragged_frac=0.5 if kind == "mixed" else 0.0)
        if kind == "empty":
            d["len1"][:] = 0
            d["len2"][:] = 0
        if kind == "len1":
            d["len1"][:] = 1
            d["len2"][:] = 1
        if kind == "allN":
            d["seq1"][:, :150] = ord("N")
            d["seq2"][::2, :150] = ord("N")
        for k in ("seq1", "qual1", "seq2", "qual2"):
            ln = d["len1"] if k.endswith("1") else d["len2"]
            d[k][np.arange(d[k].shape[1])[None, :] >= ln[:, None]] = 0
        ro, rg = o.process(*_args(d, True)), g.process(*_args(d, True))
        for k in range(3):
            assert ro[k].tobytes() == rg[k].tobytes(), (kind, k)
        assert np.array_equal(o.counters(), g.counters()), kind
    g.close()
    o.close()


def test_gpu_several_launches_and_tile_shapes(monkeypatch):
    """results must not depend on how a batch is cut into launches / tiles / workgroups"""
    paired, flags, pf, skw = cases.CASES["pe_correction"]
    d = synth.synth_pairs(30000, L=150, seed=31, **skw)
    p = pf(150)
    ref = None
    for threads, tile, cap in ((512, 0, 0), (256, 32, 2), (1024, 96, 1), (64, 8, 0)):
        monkeypatch.setenv("FASTP_GPU_THREADS", str(threads))
        if tile:
            monkeypatch.setenv("FASTP_GPU_TILE", str(tile))
        else:
            monkeypatch.delenv("FASTP_GPU_TILE", raising=False)
        if cap:
            monkeypatch.setenv("FASTP_GPU_MAX_TILES_PER_BLOCK", str(cap))
        else:
            monkeypatch.delenv("FASTP_GPU_MAX_TILES_PER_BLOCK", raising=False)
        g = engines.gpu_engine(p)
        r = g.process(*_args(d, True))
        c = g.counters()
        g.close()
        cur = (r[0].tobytes(), r[1].tobytes(), r[2].tobytes(), np.sort(r[3], order=["read", "pos"]).tobytes(), c.tobytes())
        if ref is None:
            ref = cur
        assert cur == ref, (threads, tile, cap)


def test_gpu_streamed_batches_equal_one_batch():
    """the engine is a stream processor: cutting the input into packs changes nothing
    (Stats are sums; Duplicate keeps the reference's sequential semantics across submits)"""
    p = abi.default_params(True, 150)
    d = synth.synth_pairs(24000, L=150, seed=8, dup_frac=0.3)
    g1 = engines.gpu_engine(p)
    whole = g1.process(*_args(d, True))
    c1 = g1.counters()
    g1.close()
    g2 = engines.gpu_engine(p)
    parts = []
    for a in range(0, 24000, 5000):
        sl = {k: v[a:a + 5000] for k, v in d.items()}
        parts.append(g2.process(*_args(sl, True)))
    c2 = g2.counters()
    g2.close()
    for k in range(3):
        assert whole[k].tobytes() == np.concatenate([p_[k] for p_ in parts]).tobytes(), k
    assert np.array_equal(c1, c2)


def test_gpu_full_size_properties():
    """size-independent properties on device-resident batches at bench scale (default 8M pairs;
    FASTP_FULLSIZE_PAIRS=100000000 runs BASELINE.json's configs[2] size):
      * conservation: every pair lands in exactly one filter bin, pre-stats see every read,
        post-stats see exactly the passing pairs, dup_total counts every pair
      * linearity: feeding the same batch twice doubles every additive counter
      * determinism: result records of the second pass equal the first (except the dup flag)"""
    import sys
    import torch
    sys.path.insert(0, os.path.join(engines.ROOT, "tools"))
    import synth_torch
    total = int(os.environ.get("FASTP_FULLSIZE_PAIRS", str(8 * 1024 * 1024)))
    chunk = 2 * 1024 * 1024
    p = abi.default_params(True, 150)
    p.cut_right = 1
    g = engines.gpu_engine(p)
    lay = g.layout
    dev = torch.device("cuda", 0)
    done = 0
    seed = 0
    passed_pairs = 0
    both_alive = 0
    while done < total:
        n = min(chunk, total - done)
        d = synth_torch.synth_pairs_torch(n, L=150, seed=1000 + seed, device=dev)
        s1, q1, l1 = synth_torch.pack_torch(d["seq1"], d["qual1"], d["len1"], 150)
        s2, q2, l2 = synth_torch.pack_torch(d["seq2"], d["qual2"], d["len2"], 150)
        del d
        torch.cuda.synchronize(dev)   # the batch must be complete before it is handed to the engine's stream
        outs = []
        for rep in range(2 if done == 0 else 1):
            r1 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
            r2 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
            pr = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
            nc = torch.zeros(1, dtype=torch.int32, device=dev)
            b = abi.Batch()
            b.n, b.flags = n, abi.BATCH_STAT_ISIZE
            b.seq1, b.qual1, b.len1 = s1.data_ptr(), q1.data_ptr(), l1.data_ptr()
            b.seq2, b.qual2, b.len2 = s2.data_ptr(), q2.data_ptr(), l2.data_ptr()
            res = abi.Results()
            res.r1, res.r2, res.pair = r1.data_ptr(), r2.data_ptr(), pr.data_ptr()
            res.corrections, res.corrections_capacity, res.n_corrections = None, 0, nc.data_ptr()
            if done == 0 and rep == 0:
                before = g.counters()
            g.submit_device(b, res)
            g.synchronize()
            torch.cuda.synchronize(dev)
            outs.append((r1.cpu().numpy().view(abi.READ_RESULT_DTYPE), r2.cpu().numpy().view(abi.READ_RESULT_DTYPE),
                         pr.cpu().numpy().view(abi.PAIR_RESULT_DTYPE)))
            if done == 0 and rep == 0:
                first = g.counters() - before
        if done == 0:
            second = g.counters() - before - first
            add = np.ones(lay.total, dtype=bool)
            add[:4] = False
            add[lay.dup_count] = False   # the second pass finds every pair already seen
            assert np.array_equal(first[add], second[add]), "counters are not linear in the input"
            assert second[lay.dup_count] == n, "second pass of the same batch must be 100% duplicates"
            a, b2 = outs
            for k in range(3):
                x, y = a[k].copy(), b2[k].copy()
                if k < 2:
                    x["flags"] &= ~np.uint8(abi.RF_DUP)
                    y["flags"] &= ~np.uint8(abi.RF_DUP)
                assert x.tobytes() == y.tobytes(), "results are not deterministic"
            assert (b2[0]["flags"] & abi.RF_DUP).all()
            passes = 2
        else:
            passes = 1
        r1h, r2h, _ = outs[-1]
        ok = (r1h["code"] == 0) & (r2h["code"] == 0)
        passed_pairs += int(ok.sum()) * passes
        # statInsertSize only sees pairs whose mates both survive trimAndCut (peprocessor.cpp:449)
        alive = ((r1h["flags"] | r2h["flags"]) & abi.RF_NULL) == 0
        both_alive += int(alive.sum()) * passes
        done += n
        seed += 1
    ctr = g.counters()
    g.close()
    fed = total + min(chunk, total)   # the first chunk was fed twice
    fs = ctr[lay.filter_stats: lay.filter_stats + 32]
    assert fs.sum() == 2 * fed
    assert fs[abi.PASS_FILTER] == 2 * passed_pairs
    for slot in (abi.STATS_PRE1, abi.STATS_PRE2):
        assert ctr[lay.stats[slot] + lay.st_reads] == fed
        assert ctr[lay.stats[slot] + lay.st_length_sum] == fed * 150
    for slot in (abi.STATS_POST1, abi.STATS_POST2):
        assert ctr[lay.stats[slot] + lay.st_reads] == passed_pairs
    assert ctr[lay.dup_total] == fed
    assert ctr[lay.isize: lay.isize + 513].sum() == both_alive
    for slot in range(4):
        base = lay.stats[slot]
        cyc = ctr[base + lay.st_cycle: base + lay.st_cycle + 34 * 150].reshape(34, 150)
        assert np.array_equal(cyc[16:24].sum(0), cyc[32]), "per-base contents must add up to total bases"
        assert np.array_equal(cyc[24:32].sum(0), cyc[33])
        assert cyc[32].sum() == ctr[base + lay.st_length_sum]
        assert ctr[base + lay.st_qual_hist: base + lay.st_qual_hist + 128].sum() == cyc[32].sum()


@pytest.mark.parametrize("name", cases.GAP_CASES)
def test_gpu_one_gap_accept_paths(name):
    """Matcher::diffWithOneInsertion / matchWithOneInsertion ACCEPT on these inputs: the oracle's positives are
    counted first (so the comparison cannot pass on "both said no"), then every record and counter is compared"""
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(20000, L=150, seed=177, paired=paired, **skw)
    params = pf(150)
    o = oraclelib.Oracle(params)
    ro = o.process(*_args(d, paired))
    o.close()
    if "allow_gap" in name:
        tot, neg = gap_util.gap_overlap_pairs(d, params, limit=6000)
        assert tot > 100 and neg > 100, f"{name}: only {tot} one-gap overlaps ({neg} with a negative offset) in 6000 pairs"
        assert gap_util.gap_trimmed_pairs(ro[0], ro[2]) > 100
    else:
        assert gap_util.gap_adapter_trims(d["seq1"], d["len1"], ro[0], bytes(params.adapter_seq_r1)) > 1000
        if paired:
            assert gap_util.gap_adapter_trims(d["seq2"], d["len2"], ro[1], bytes(params.adapter_seq_r2)) > 1000
    _compare(name, params, d, paired)


def test_gpu_equals_oracle_at_baseline_scale():
    """BASELINE configs[2] options (auto-adapter by overlap + --cut_right, duplicate evaluation, default filters)
    on >= 4 M synthetic 2x150 pairs against the ORACLE - every record, every counter, every duplicate decision -
    so that the driver-run suite carries parity at bench scale, not only properties.  The per-read part of the
    oracle runs on contiguous chunks in parallel threads (its counters add); the stream-ordered duplicate decisions
    are checked against oraclelib.sequential_duplicates over the whole stream."""
    import sys
    from concurrent.futures import ThreadPoolExecutor
    import torch
    sys.path.insert(0, os.path.join(engines.ROOT, "tools"))
    import synth_torch
    total = int(os.environ.get("FASTP_SCALE_PAIRS", str(4 * 1024 * 1024)))
    chunk = 256 * 1024
    p = abi.default_params(True, 150)
    p.cut_right = 1
    dev = torch.device("cuda", 0)
    g = engines.gpu_engine(p)
    parts, recs = [], []
    for k, start in enumerate(range(0, total, 1024 * 1024)):
        n = min(1024 * 1024, total - start)
        d = synth_torch.synth_pairs_torch(n, L=150, seed=7000 + k, device=dev)
        s1, q1, l1 = synth_torch.pack_torch(d["seq1"], d["qual1"], d["len1"], 150)
        s2, q2, l2 = synth_torch.pack_torch(d["seq2"], d["qual2"], d["len2"], 150)
        torch.cuda.synchronize(dev)
        r1 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
        r2 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
        pr = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
        nc = torch.zeros(1, dtype=torch.int32, device=dev)
        b = abi.Batch()
        b.n, b.flags = n, abi.BATCH_STAT_ISIZE
        b.seq1, b.qual1, b.len1 = s1.data_ptr(), q1.data_ptr(), l1.data_ptr()
        b.seq2, b.qual2, b.len2 = s2.data_ptr(), q2.data_ptr(), l2.data_ptr()
        res = abi.Results()
        res.r1, res.r2, res.pair = r1.data_ptr(), r2.data_ptr(), pr.data_ptr()
        res.corrections, res.corrections_capacity, res.n_corrections = None, 0, nc.data_ptr()
        g.submit_device(b, res)
        g.synchronize()
        torch.cuda.synchronize(dev)
        recs.append((r1.cpu().numpy().view(abi.READ_RESULT_DTYPE), r2.cpu().numpy().view(abi.READ_RESULT_DTYPE),
                     pr.cpu().numpy().view(abi.PAIR_RESULT_DTYPE)))
        pad = lambda a: np.pad(a.cpu().numpy(), ((0, 0), (0, 2)))
        parts.append({kk: (pad(d[kk]) if kk[0] in "sq" else d[kk].cpu().numpy().astype(np.int32)) for kk in
                      ("seq1", "qual1", "len1", "seq2", "qual2", "len2")})
        del d, s1, q1, l1, s2, q2, l2
    cg = g.counters()
    lay = g.layout
    g.close()
    rg = [np.concatenate([r[k] for r in recs]) for k in range(3)]
    full = {kk: np.concatenate([pt[kk] for pt in parts]) for kk in parts[0]}
    del parts, recs

    def oracle_chunk(lo):
        hi = min(total, lo + chunk)
        o = oraclelib.Oracle(p)
        r = o.process(full["seq1"][lo:hi], full["qual1"][lo:hi], full["len1"][lo:hi], full["seq2"][lo:hi],
                      full["qual2"][lo:hi], full["len2"][lo:hi], corr_capacity=16)
        c = o.counters()
        o.close()
        return r[:3], c

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        outs = list(ex.map(oracle_chunk, range(0, total, chunk)))
    co = np.zeros_like(cg)
    for _, c in outs:
        co += c
    co[:4] = outs[0][1][:4]          # header words (ABI version, cycles, ...) are not additive
    dup = oraclelib.sequential_duplicates(p.dup_accuracy_level, full["seq1"], full["len1"], full["seq2"], full["len2"])
    for k, what in enumerate(("read1 results", "read2 results", "pair results")):
        ro = np.concatenate([o[0][k] for o in outs])
        if k < 2:   # the chunk oracles each started with an empty bloom filter: take the stream's decision
            ro["flags"] = (ro["flags"] & ~np.uint8(abi.RF_DUP)) | np.where(dup, abi.RF_DUP, 0).astype(np.uint8)
        bad = np.nonzero(ro != rg[k])[0]
        assert len(bad) == 0, f"{what} differ at {len(bad)} of {total}, first {bad[:5]}: oracle {ro[bad[:3]]} gpu {rg[k][bad[:3]]}"
    co[lay.dup_count] = int(dup.sum())
    bad = np.nonzero(co != cg)[0]
    assert len(bad) == 0, f"{len(bad)} counters differ, first at {bad[:8]}: oracle {co[bad[:8]]} gpu {cg[bad[:8]]}"
    assert cg[lay.dup_total] == total and dup.sum() > 1000   # exact copies only (most synthetic duplicates differ by an error)


@pytest.mark.parametrize("level,L,paired", [(1, 150, True), (3, 150, True), (1, 37, True), (1, 250, True), (1, 150, False), (4, 100, True), (6, 100, True)])
def test_gpu_duplicate_hash_bit_positions_equal_oracle(level, L, paired):
    """the hash itself (Duplicate::seq2intvector mod mBufLenInBits), not only the decisions it leads to"""
    import torch
    import shard_util
    p = abi.default_params(paired, L)
    p.dup_accuracy_level = level
    if not paired:
        p.adapter_seq_r1 = None
    d = synth.synth_pairs(20000, L=L, seed=3 + level, paired=paired, ragged_frac=0.6, insert_mean=L * 1.2, insert_sd=L * 0.4)
    g = engines.gpu_engine(p)
    got = shard_util.device_bit_positions(g, d, torch.device("cuda", 0))
    g.close()
    want = shard_util.oracle_bit_positions(level, d)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} bit positions differ"


def test_gpu_reset_starts_a_new_run():
    """fastp_gpu_reset: counters, bloom bitmaps and stream positions as after create"""
    p = abi.default_params(True, 150)
    p.cut_right = 1
    d = synth.synth_pairs(9000, L=150, seed=12, dup_frac=0.3)
    g = engines.gpu_engine(p)
    first = g.process(*_args(d, True))
    c1 = g.counters()
    g.reset()
    z = g.counters()
    assert not z[4:].any() and np.array_equal(z[:4], c1[:4])
    again = g.process(*_args(d, True))
    c2 = g.counters()
    g.close()
    assert np.array_equal(c1, c2)
    for a, b in zip(first[:3], again[:3]):
        assert a.tobytes() == b.tobytes()      # incl. RF_DUP: the bloom filter started empty again


def test_gpu_cabi_rccl_allreduce_single_rank():
    """fastp_gpu_comm_id / comm_init / allreduce / exchange_dup_prefix through librccl with a one-rank
    communicator (this box has one GPU; RCCL refuses two ranks on one device): the sum over one rank is the
    block itself, header words intact, and the engine keeps working afterwards"""
    p = abi.default_params(True, 150)
    d = synth.synth_pairs(5000, L=150, seed=13)
    g = engines.gpu_engine(p)
    g.process(*_args(d, True))
    before = g.counters()
    g.comm_init(g.comm_id(), 1, 0)
    g.allreduce()
    assert np.array_equal(g.counters(), before)
    g.exchange_dup_prefix()
    g.allreduce()
    assert np.array_equal(g.counters(), before)
    g.process(*_args(d, True))
    after = g.counters()
    lay = g.layout
    assert after[lay.dup_total] == 2 * before[lay.dup_total]
    g.close()


def test_gpu_counter_export_import_merge_rehearsal():
    """the device side of the multi-GPU merge on one GPU: two engines take the two shards, their
    counter blocks are exported into torch tensors, summed (what the RCCL all-reduce does) and
    imported back - the merged block equals a single engine's that saw everything"""
    import torch
    from fastp_amd import multigpu
    p = abi.default_params(True, 150)
    p.cut_right = 1
    d = synth.synth_pairs(30000, L=150, seed=11)
    dev = torch.device("cuda", 0)
    whole = engines.gpu_engine(p)
    whole.process(*_args(d, True))
    cw = whole.counters()
    lay = whole.layout
    whole.close()
    engs, bufs = [], []
    for rank in range(2):
        lo, hi = multigpu.shard_bounds(30000, 2, rank)
        e = engines.gpu_engine(p)
        e.process(*_args({k: v[lo:hi] for k, v in d.items()}, True))
        t = torch.empty(lay.total, dtype=torch.int64, device=dev)
        e.counters_export(t.data_ptr())
        engs.append(e)
        bufs.append(t)
    merged = bufs[0] + bufs[1]
    torch.cuda.synchronize(dev)
    for e in engs:
        e.counters_import(merged.data_ptr())
    c0, c1 = engs[0].counters(), engs[1].counters()
    for e in engs:
        e.close()
    assert np.array_equal(c0, c1)
    keep = np.ones(lay.total, dtype=bool)
    keep[lay.dup_count] = False   # cross-shard duplicates are not seen (per-shard bitmaps)
    assert np.array_equal(c0[keep], cw[keep])
    assert c0[lay.dup_count] <= cw[lay.dup_count]


@pytest.mark.parametrize("eol", [b"\n", b"\r\n"])
def test_gpu_device_fastq_parse_feeds_the_engine(eol):
    """FASTQ text in HBM -> fastp_gpu_parse_fastq -> fastp_gpu_submit_device on the parsed rows: same packed
    rows as FastqReader's line splitting + the host packer, same records and counters as the host-packed path"""
    import torch
    import parse_util
    dev = torch.device("cuda", 0)
    p = abi.default_params(True, 150)
    p.cut_right = 1
    n = 20000
    d = synth.synth_pairs(n, L=150, seed=31)
    ref = engines.gpu_engine(p)
    want = ref.process(*_args(d, True))
    cref = ref.counters()
    ref.close()
    g = engines.gpu_engine(p)
    ss, qs = abi.seq_stride(150), abi.qual_stride(150)
    packed = []
    for mate in (1, 2):
        txt = synth.to_fastq(d[f"seq{mate}"], d[f"qual{mate}"], d[f"len{mate}"], mate).replace(b"\n", eol)
        exp = parse_util.expected(txt, 150, None, True)
        pad = (-len(txt)) % 16 + 16
        t = torch.frombuffer(bytearray(txt + b"\0" * pad), dtype=torch.uint8).to(dev)
        assert t.data_ptr() % 16 == 0
        seq = torch.full((n, ss), 0xEE, dtype=torch.uint8, device=dev)
        qual = torch.full((n, qs), 0xEE, dtype=torch.uint8, device=dev)
        lens = torch.zeros(n, dtype=torch.int16, device=dev)
        loff = torch.zeros(4 * n, dtype=torch.int32, device=dev)
        llen = torch.zeros(4 * n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        info = g.parse_fastq(t.data_ptr(), len(txt), True, n, seq.data_ptr(), qual.data_ptr(), lens.data_ptr(),
                             loff.data_ptr(), llen.data_ptr())
        assert info.n_records == n and info.first_bad == -1 and info.consumed == len(txt)
        assert np.array_equal(seq.cpu().numpy(), exp[0]) and np.array_equal(qual.cpu().numpy(), exp[1])
        assert np.array_equal(lens.cpu().numpy().view(np.uint16), exp[2])
        assert np.array_equal(loff.cpu().numpy().view(np.uint32), exp[3])
        assert np.array_equal(llen.cpu().numpy().view(np.uint32), exp[4])
        packed.append((seq, qual, lens))
    r1 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
    r2 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
    pr = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
    nc = torch.zeros(1, dtype=torch.int32, device=dev)
    b = abi.Batch()
    b.n, b.flags = n, abi.BATCH_STAT_ISIZE
    b.seq1, b.qual1, b.len1 = (x.data_ptr() for x in packed[0])
    b.seq2, b.qual2, b.len2 = (x.data_ptr() for x in packed[1])
    res = abi.Results()
    res.r1, res.r2, res.pair = r1.data_ptr(), r2.data_ptr(), pr.data_ptr()
    res.corrections, res.corrections_capacity, res.n_corrections = None, 0, nc.data_ptr()
    torch.cuda.synchronize(dev)
    g.submit_device(b, res)
    g.synchronize()
    assert r1.cpu().numpy().view(abi.READ_RESULT_DTYPE).tobytes() == want[0].tobytes()
    assert r2.cpu().numpy().view(abi.READ_RESULT_DTYPE).tobytes() == want[1].tobytes()
    assert pr.cpu().numpy().view(abi.PAIR_RESULT_DTYPE).tobytes() == want[2].tobytes()
    assert np.array_equal(g.counters(), cref)
    g.close()


@pytest.mark.parametrize("name", ["pe_default", "pe_correction", "pe_filters", "pe_noadapter_dedup", "pe_adapter_fasta",
                                  "se_adapter_cut", "se_polyx_complexity"])
def test_gpu_device_fastq_format_equals_host_writer(name):
    """text in HBM -> parse -> submit_device -> fastp_gpu_format_fastq: out1/out2 text == the host writer's"""
    import format_util
    import test_hostsim_parity as hs
    want = hs._format_case(engines.gpu_engine, format_util.TorchMem(), name, 12000)
    assert len(want.out1) > 0


@pytest.mark.parametrize("name,want_failed,want_unpaired", [
    ("pe_filters", True, True), ("pe_filters", True, False), ("pe_merge", True, False), ("pe_merge_unmerged", True, False),
    ("pe_umi_per_read", True, True), ("se_umi_read1", True, False), ("pe_correction", True, True),
    ("pe_noadapter_dedup", True, True), ("se_polyx_complexity", True, False)])
def test_gpu_device_all_streams_equal_host_writer(name, want_failed, want_unpaired):
    """fastp_gpu_format_streams: out1/out2/failed/merged/unpaired text assembled in HBM == the host writer's"""
    import format_util
    import test_hostsim_parity as hs
    got = hs._streams_case(engines.gpu_engine, format_util.TorchMem(), name, 20000, want_failed, want_unpaired)
    assert sum(len(v) for v in got.values()) > 0
    if "merge" in name:
        assert b" merged_" in got["merged"]


def test_gpu_device_all_streams_crlf_prefix_overflow():
    import format_util
    import test_hostsim_parity as hs
    got = hs._streams_case(engines.gpu_engine, format_util.TorchMem(), "pe_umi_per_read", 3000, True, False, umi_extra=(b"UMI", b"-"))
    assert b"-UMI_" in got["out1"]
    hs._streams_case(engines.gpu_engine, format_util.TorchMem(), "pe_merge", 3000, True, False, eol=b"\r\n")
    paired, flags, pf, skw = cases.CASES["pe_filters"]
    d = synth.synth_pairs(2000, L=150, seed=5)
    p = cases.finalize_params("pe_filters", pf(150), d["seq1"], d["len1"], d["seq2"], d["len2"])
    g = engines.gpu_engine(p)
    rc, got, lens = format_util.run_streams(g, format_util.TorchMem(), p, synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1),
                                            synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2), 150, shrink=1)
    assert rc == abi.E_OVERFLOW and lens[1] > 100 and lens[0] == len(got["out1"])
    g.close()


def test_gpu_evaluator_prepass_equals_port():
    """the Evaluator's counting loops on the device (fastp_gpu_eval_*) at the reference's own sample sizes: the
    1.51 Mbase limit of computeOverRepSeq cuts inside the batch; ten-mer histogram over all reads"""
    import format_util
    import test_hostsim_parity as hs
    got, wc = hs._eval_case(engines.gpu_engine, format_util.TorchMem(), 12000, 21)
    assert len(got) >= 3
    hs._eval_case(engines.gpu_engine, format_util.TorchMem(), 3000, 22, L=102, trim_tail1=2)


def test_gpu_deflate_bgzf_members_round_trip():
    """fastp_gpu_deflate_bgzf on 20 MB of FASTQ text, 3 MB of noise, runs and block-boundary sizes: gzip, and our own
    BGZF index + inflate with CRC check, return the text"""
    import format_util
    import test_hostsim_parity as hs
    hs._deflate_case(engines.gpu_engine, format_util.TorchMem(), big=True)


def test_gpu_device_fastq_format_crlf_and_overflow():
    import format_util
    import test_hostsim_parity as hs
    hs._format_case(engines.gpu_engine, format_util.TorchMem(), "pe_correction", 3000, eol=b"\r\n")
    d = synth.synth_pairs(1000, L=150, seed=5, paired=False)
    fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    p = abi.default_params(False, 150)
    g = engines.gpu_engine(p)
    rc, o1, _, lens = format_util.run(g, format_util.TorchMem(), p, fq1, None, 150, out_slack=-(len(fq1) // 2))
    assert rc == abi.E_OVERFLOW and lens[0] > len(o1)
    g.close()


@pytest.mark.parametrize("name,npacks", [("pe_default", 2), ("pe_noadapter_dedup", 3), ("pe_overrep", 2)])
def test_gpu_two_shard_exact_protocol_equals_one_stream(name, npacks):
    """two ranks (gloo, both on cuda:0) through multigpu.run_shard: records and counters - duplicates across the
    shard boundary and overrepresentation sampling positions included - equal ONE stream (the oracle's)"""
    import os
    import torch.multiprocessing as mp
    import oraclelib
    import shard_util
    n = 30000
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + 13
    mp.spawn(shard_util.shard_worker, args=(2, port, ret, "gpu", name, n, npacks, 150), nprocs=2, join=True)
    params, d, paired = shard_util.case_input(name, n, 150)
    o = oraclelib.Oracle(params)
    whole = o.process(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    ctr, lay = o.counters(), o.layout
    o.close()
    assert np.array_equal(ret[0][0], ret[1][0])
    assert ctr[lay.dup_count] > 0
    bad = np.nonzero(ret[0][0] != ctr)[0]
    assert len(bad) == 0, f"{name}: counters differ at {bad[:8]}"
    for k in range(3):
        assert ret[0][k + 1] + ret[1][k + 1] == whole[k].tobytes(), f"{name}: records {k} differ"


@pytest.mark.parametrize("name", ["pe_default", "pe_correction", "se_adapter_cut"])
def test_gpu_file_pipeline_equals_reference_outputs(name, tmp_path):
    """FASTQ files -> fastp_amd.pipeline (parse, worker loop, format on the device; tiny chunks so that records
    straddle chunk borders) -> out1/out2 files: byte-identical to the host writer's and, where the reference
    binary travelled to this box, to reference fastp's own output files; counters == the one-stream engine's"""
    from fastp_amd import pipeline
    paired, flags, pf, skw = cases.CASES[name]
    n = 20500     # not a multiple of max_records: the last trip is a short one
    d = synth.synth_pairs(n, L=150, seed=91, paired=paired, **skw)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    fq2 = synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2) if paired else None
    ref = engines.gpu_engine(params)
    want, ctr, _ = driver.run_engine(ref, params, fq1, fq2, pack=n, stride=abi.qual_stride(150))
    ref.close()
    (tmp_path / "r1.fq").write_bytes(fq1)
    if paired:
        (tmp_path / "r2.fq").write_bytes(fq2)
    pl = pipeline.FastqPipeline(params, chunk_bytes=1 << 20, max_records=2500)
    st = pl.run(str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq") if paired else None, str(tmp_path / "o1.fq"),
                str(tmp_path / "o2.fq") if paired else None)
    got_ctr = pl.counters()
    pl.close()
    assert st["units"] == n and st["chunks"] > 3
    assert (tmp_path / "o1.fq").read_bytes() == bytes(want.out1)
    if paired:
        assert (tmp_path / "o2.fq").read_bytes() == bytes(want.out2)
    assert np.array_equal(got_ctr, ctr)
    if driver.have_reference_binary():
        r = driver.run_reference(flags, fq1, fq2)
        assert (tmp_path / "o1.fq").read_bytes() == r["out1"]
        if paired:
            assert (tmp_path / "o2.fq").read_bytes() == r["out2"]


def test_gpu_file_pipeline_drains_records_left_at_eof(tmp_path):
    """the whole file fits ONE chunk but holds several times max_records records: after EOF the pipeline must keep
    parsing the carried text (no refill) instead of stopping after the first max_records"""
    from fastp_amd import pipeline
    n = 2300
    p = abi.default_params(True, 150)
    d = synth.synth_pairs(n, L=150, seed=92)
    fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    fq2 = synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2)
    ref = engines.gpu_engine(p)
    want, ctr, _ = driver.run_engine(ref, p, fq1, fq2, pack=n, stride=abi.qual_stride(150))
    ref.close()
    (tmp_path / "r1.fq").write_bytes(fq1)
    (tmp_path / "r2.fq").write_bytes(fq2)
    pl = pipeline.FastqPipeline(p, chunk_bytes=4 << 20, max_records=700)
    st = pl.run(str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq"))
    got_ctr = pl.counters()
    pl.close()
    assert st["units"] == n and st["chunks"] == 4
    assert (tmp_path / "o1.fq").read_bytes() == bytes(want.out1)
    assert (tmp_path / "o2.fq").read_bytes() == bytes(want.out2)
    assert np.array_equal(got_ctr, ctr)


@pytest.mark.parametrize("level,strategy", [(6, "default"), (1, "default"), (0, "default"), (6, "fixed")])
def test_gpu_bgzf_inflate_equals_zlib(level, strategy):
    """BGZF blocks inflated on the device (one lane per block) == the text zlib compressed; then parsed and packed"""
    import zlib
    import bgzf_util
    import format_util
    import test_hostsim_parity as hs
    strat = {"default": zlib.Z_DEFAULT_STRATEGY, "fixed": zlib.Z_FIXED}[strategy]
    text = hs._se_fastq_text(30000, 8)
    comp = bgzf_util.compress(text, level=level, strategy=strat)
    g = engines.gpu_engine(abi.default_params(False, 150))
    info, rc, bad, got = hs._inflate(g, format_util.TorchMem(), comp)
    assert rc == 0 and bad == -1 and info.consumed == len(comp) and info.n_blocks == len(text) // 0xff00 + 2
    assert got == text
    bad_comp = bytearray(comp)
    bad_comp[18 + 3000] ^= 0x10
    _, rc, bad, _ = hs._inflate(g, format_util.TorchMem(), bytes(bad_comp), check=False)
    assert rc == abi.E_INVALID and bad == 0
    g.close()


@pytest.mark.parametrize("variant", ["wave", "lane"])
def test_gpu_bgzf_inflate_survives_corruption(variant, monkeypatch):
    """120 corrupted chunks through either inflate kernel: errors or (CRC off) different text, never a write outside
    the output, never an accept with the CRC check on, and the GPU stays alive"""
    import format_util
    import test_hostsim_parity as hs
    monkeypatch.setenv("FASTP_GPU_INFLATE", variant)
    hs._corruption_case(engines.gpu_engine, format_util.TorchMem(), 3000, 0xff00, 120, 60)


def test_gpu_file_pipeline_reads_bgzf(tmp_path):
    """the same pipeline fed BGZF-compressed inputs (inflated on the device): same output files, same counters"""
    import bgzf_util
    from fastp_amd import pipeline
    name = "pe_correction"
    paired, flags, pf, skw = cases.CASES[name]
    n = 20000
    d = synth.synth_pairs(n, L=150, seed=92, paired=True, **skw)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d["seq2"], d["len2"])
    fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    fq2 = synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2)
    ref = engines.gpu_engine(params)
    want, ctr, _ = driver.run_engine(ref, params, fq1, fq2, pack=n, stride=abi.qual_stride(150))
    ref.close()
    (tmp_path / "r1.fq.gz").write_bytes(bgzf_util.compress(fq1))
    (tmp_path / "r2.fq").write_bytes(fq2)                      # one compressed, one plain
    assert pipeline.FastqPipeline.is_bgzf(str(tmp_path / "r1.fq.gz")) and not pipeline.FastqPipeline.is_bgzf(str(tmp_path / "r2.fq"))
    pl = pipeline.FastqPipeline(params, chunk_bytes=1 << 20)
    st = pl.run(str(tmp_path / "r1.fq.gz"), str(tmp_path / "r2.fq"), str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq"))
    got_ctr = pl.counters()
    pl.close()
    assert st["units"] == n and st["chunks"] > 3
    assert (tmp_path / "o1.fq").read_bytes() == bytes(want.out1)
    assert (tmp_path / "o2.fq").read_bytes() == bytes(want.out2)
    assert np.array_equal(got_ctr, ctr)
    import gzip
    (tmp_path / "plain.gz").write_bytes(gzip.compress(fq1[:5000]))
    pl = pipeline.FastqPipeline(params, chunk_bytes=1 << 20)
    with pytest.raises(pipeline.PipelineError):
        pl.run(str(tmp_path / "plain.gz"), str(tmp_path / "r2.fq"), str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq"))
    pl.close()


@pytest.mark.parametrize("name", ["pe_filters", "pe_merge", "pe_umi_per_read", "se_umi_read1"])
def test_gpu_file_pipeline_all_streams_and_gzip_outputs(name, tmp_path):
    """files in (BGZF) -> every output stream assembled and gzip-compressed on the device -> .gz files: their text ==
    the host writer's streams (and reference fastp's files where the binary is present); the .gz files are valid BGZF"""
    import gzip
    import bgzf_util
    from fastp_amd import hostloop, pipeline
    paired, flags, pf, skw = cases.CASES[name]
    n = 15000
    d = synth.synth_pairs(n, L=150, seed=93, paired=paired, **skw)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    fq2 = synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2) if paired else None
    umi = cases.UMI.get(name)
    editor = hostloop.UmiNameEditor(*umi) if umi else None
    ref = engines.gpu_engine(params)
    want, ctr, _ = driver.run_engine(ref, params, fq1, fq2, pack=n, stride=abi.qual_stride(150), want_failed=True,
                                     want_unpaired=paired, umi=editor)
    ref.close()
    (tmp_path / "r1.fq.gz").write_bytes(bgzf_util.compress(fq1))
    if paired:
        (tmp_path / "r2.fq.gz").write_bytes(bgzf_util.compress(fq2))
    t = lambda f: str(tmp_path / f)
    pl = pipeline.FastqPipeline(params, chunk_bytes=1 << 20, max_records=3000)
    st = pl.run(t("r1.fq.gz"), t("r2.fq.gz") if paired else None, t("o1.fq.gz"), t("o2.fq.gz") if paired else None,
                failed_out=t("failed.fq.gz"), merged_out=t("merged.fq.gz") if params.merge else None,
                unpaired1=t("u1.fq") if paired else None, unpaired2=t("u2.fq.gz") if paired else None, umi=umi)
    got_ctr = pl.counters()
    pl.close()
    assert st["units"] == n and st["chunks"] > 3
    assert np.array_equal(got_ctr, ctr)

    def text(fn):
        raw = (tmp_path / fn).read_bytes()
        if fn.endswith(".gz"):
            assert pipeline.FastqPipeline.is_bgzf(t(fn)) and raw.endswith(pipeline.FastqPipeline.EOF_MEMBER)
            return gzip.decompress(raw)
        return raw
    assert text("o1.fq.gz") == bytes(want.out1)
    assert text("failed.fq.gz") == bytes(want.failed)
    if paired:
        assert text("o2.fq.gz") == bytes(want.out2)
        assert text("u1.fq") == bytes(want.unpaired1) and text("u2.fq.gz") == bytes(want.unpaired2)
    if params.merge:
        assert text("merged.fq.gz") == bytes(want.merged) and len(want.merged) > 0
    if driver.have_reference_binary() and name != "pe_filters":   # (run_reference does not ask for the unpaired files)
        r = driver.run_reference(flags, fq1, fq2)
        assert text("o1.fq.gz") == (r["out1"] or b"")
        if params.merge:
            assert text("merged.fq.gz") == r["merged"]


def test_gpu_missing_library_fails_loudly(tmp_path):
    with pytest.raises(FileNotFoundError):
        engine.load_library(str(tmp_path / "nope.so"))
def _se_scale_against_oracle(p, expect_plan, check):
    """2 Mi synthetic single-end reads against the ORACLE: every record, every counter, every duplicate decision (chunked
    oracle for the per-read part, oraclelib.sequential_duplicates for the stream-ordered part - the harness of the
    configs[2] test)."""
    import sys
    from concurrent.futures import ThreadPoolExecutor
    import torch
    sys.path.insert(0, os.path.join(engines.ROOT, "tools"))
    import synth_torch
    total = int(os.environ.get("FASTP_SCALE_READS", str(2 * 1024 * 1024)))
    chunk = 256 * 1024
    dev = torch.device("cuda", 0)
    g = engines.gpu_engine(p)
    parts, recs = [], []
    for k, start in enumerate(range(0, total, 1024 * 1024)):
        n = min(1024 * 1024, total - start)
        d = synth_torch.synth_pairs_torch(n, L=150, seed=7100 + k, device=dev)
        s1, q1, l1 = synth_torch.pack_torch(d["seq1"], d["qual1"], d["len1"], 150)
        r1 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
        nc = torch.zeros(1, dtype=torch.int32, device=dev)
        b = abi.Batch()
        b.n, b.flags = n, 0
        b.seq1, b.qual1, b.len1 = s1.data_ptr(), q1.data_ptr(), l1.data_ptr()
        res = abi.Results()
        res.r1 = r1.data_ptr()
        res.corrections, res.corrections_capacity, res.n_corrections = None, 0, nc.data_ptr()
        torch.cuda.synchronize(dev)
        g.submit_device(b, res)
        g.synchronize()
        recs.append(r1.cpu().numpy().view(abi.READ_RESULT_DTYPE))
        pad = lambda a: np.pad(a.cpu().numpy(), ((0, 0), (0, 2)))
        parts.append({"seq1": pad(d["seq1"]), "qual1": pad(d["qual1"]), "len1": d["len1"].cpu().numpy().astype(np.int32)})
        del d, s1, q1, l1
    assert g.plan() == expect_plan
    cg = g.counters()
    lay = g.layout
    g.close()
    rg = np.concatenate(recs)
    full = {kk: np.concatenate([pt[kk] for pt in parts]) for kk in parts[0]}
    del parts, recs

    def oracle_chunk(lo):
        hi = min(total, lo + chunk)
        o = oraclelib.Oracle(p)
        r = o.process(full["seq1"][lo:hi], full["qual1"][lo:hi], full["len1"][lo:hi], corr_capacity=16)
        c = o.counters()
        o.close()
        return r[0], c

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        outs = list(ex.map(oracle_chunk, range(0, total, chunk)))
    co = np.zeros_like(cg)
    for _, c in outs:
        co += c
    co[:4] = outs[0][1][:4]
    dup = oraclelib.sequential_duplicates(p.dup_accuracy_level, full["seq1"], full["len1"], None, None)
    ro = np.concatenate([o[0] for o in outs])
    ro["flags"] = (ro["flags"] & ~np.uint8(abi.RF_DUP)) | np.where(dup, abi.RF_DUP, 0).astype(np.uint8)
    bad = np.nonzero(ro != rg)[0]
    assert len(bad) == 0, f"records differ at {len(bad)} of {total}, first {bad[:5]}: oracle {ro[bad[:3]]} gpu {rg[bad[:3]]}"
    check(rg, ro, full, total)
    co[lay.dup_count] = int(dup.sum())
    bad = np.nonzero(co != cg)[0]
    assert len(bad) == 0, f"{len(bad)} counters differ, first at {bad[:8]}: oracle {co[bad[:8]]} gpu {cg[bad[:8]]}"


@pytest.mark.gpu
def test_gpu_equals_oracle_at_scale_config1_single_end():
    """BASELINE configs[1] (SE 1x150, sliding-window quality trim + polyG only: `-A -g --cut_right`)"""
    p = abi.default_params(False, 150)
    p.adapter_seq_r1 = None
    p.adapter_enabled = 0
    p.poly_g = 1
    p.cut_right = 1

    def check(rg, ro, full, total):
        assert int((rg["flags"] & abi.RF_POLYX).sum()) == 0 and int((ro["len"] < full["len1"]).sum()) > total // 10   # the trims do happen
    _se_scale_against_oracle(p, "lane", check)


@pytest.mark.gpu
def test_gpu_equals_oracle_at_scale_single_end_default_adapter_by_sequence():
    """fastp's DEFAULT single-end run as the Evaluator leaves it: the detected adapter trimmed by sequence
    (AdapterTrimmer::trimBySequence, seprocessor.cpp:240-252) - on the lane plan since round 4 - plus polyX trimming and the
    complexity filter, which ride in the same kernel instantiation"""
    p = abi.default_params(False, 150)
    p.adapter_seq_r1 = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    p.poly_x = 1
    p.complexity_filter = 1

    def check(rg, ro, full, total):
        assert int((rg["flags"] & abi.RF_ADAPTER).sum()) > total // 20        # the adapter is found in a good share of the reads
        assert int((rg["adapter_pos"] < 0).sum()) >= 0
    _se_scale_against_oracle(p, "lane", check)


@pytest.mark.gpu
def test_gpu_equals_oracle_at_scale_config4_dedup_overrep():
    """BASELINE configs[4]'s options (PE 2x250, --dedup, overrepresentation analysis with the Evaluator's seeds) on 1 Mi
    synthetic pairs against the ORACLE run over the whole stream on one thread (the duplicate decision feeds the
    routing and the sampling positions are stream positions: nothing here can be chunked): records + every counter."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(engines.ROOT, "tools"))
    import synth_torch
    import evalport
    total = int(os.environ.get("FASTP_SCALE_PAIRS4", str(1024 * 1024)))
    L = 250
    p = abi.default_params(True, L)
    p.cut_right = 1
    p.dedup = 1
    dsmall = synth_torch.synth_pairs_torch(20000, L=L, seed=5, device="cpu")
    pad6 = lambda a: np.pad(a.numpy(), ((0, 0), (0, 6)))
    b1 = cases._ArrayBatch(pad6(dsmall["seq1"]), dsmall["len1"].numpy())
    b2 = cases._ArrayBatch(pad6(dsmall["seq2"]), dsmall["len2"].numpy())
    e1, e2 = evalport.evaluate_seq_len(b1), evalport.evaluate_seq_len(b2)
    abi.set_overrep(p, evalport.evaluate_overrep_seqs(b1, e1), evalport.evaluate_overrep_seqs(b2, e2), e1, e2, 20)
    assert p.n_overrep_seqs1 > 0
    dev = torch.device("cuda", 0)
    g = engines.gpu_engine(p)
    parts, recs = [], []
    step = 256 * 1024
    for k, start in enumerate(range(0, total, step)):
        n = min(step, total - start)
        d = synth_torch.synth_pairs_torch(n, L=L, seed=7200 + k, device=dev)
        # exact duplicates (the synthesizer's share a fragment but not their sequencing errors): a fifth of the rows
        # become copies of other rows of the batch
        gen = torch.Generator(device=dev)
        gen.manual_seed(99 + k)
        dst = torch.randperm(n, generator=gen, device=dev)[:n // 5]
        src = torch.randint(0, n, (n // 5,), generator=gen, device=dev)
        for kk in ("seq1", "qual1", "seq2", "qual2"):
            d[kk][dst] = d[kk][src]
        s1, q1, l1 = synth_torch.pack_torch(d["seq1"], d["qual1"], d["len1"], L)
        s2, q2, l2 = synth_torch.pack_torch(d["seq2"], d["qual2"], d["len2"], L)
        r1 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
        r2 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
        pr = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
        nc = torch.zeros(1, dtype=torch.int32, device=dev)
        b = abi.Batch()
        b.n, b.flags = n, abi.BATCH_STAT_ISIZE
        b.seq1, b.qual1, b.len1 = s1.data_ptr(), q1.data_ptr(), l1.data_ptr()
        b.seq2, b.qual2, b.len2 = s2.data_ptr(), q2.data_ptr(), l2.data_ptr()
        res = abi.Results()
        res.r1, res.r2, res.pair = r1.data_ptr(), r2.data_ptr(), pr.data_ptr()
        res.corrections, res.corrections_capacity, res.n_corrections = None, 0, nc.data_ptr()
        torch.cuda.synchronize(dev)
        g.submit_device(b, res)
        g.synchronize()
        recs.append((r1.cpu().numpy().view(abi.READ_RESULT_DTYPE), r2.cpu().numpy().view(abi.READ_RESULT_DTYPE),
                     pr.cpu().numpy().view(abi.PAIR_RESULT_DTYPE)))
        pad = lambda a: np.pad(a.cpu().numpy(), ((0, 0), (0, 6)))
        parts.append({kk: (pad(d[kk]) if kk[0] in "sq" else d[kk].cpu().numpy().astype(np.int32)) for kk in
                      ("seq1", "qual1", "len1", "seq2", "qual2", "len2")})
        del d, s1, q1, l1, s2, q2, l2
    cg = g.counters()
    g.close()
    rg = [np.concatenate([r[k] for r in recs]) for k in range(3)]
    full = {kk: np.concatenate([pt[kk] for pt in parts]) for kk in parts[0]}
    del parts, recs
    o = oraclelib.Oracle(p)
    ro = o.process(full["seq1"], full["qual1"], full["len1"], full["seq2"], full["qual2"], full["len2"], corr_capacity=16)
    co = o.counters()
    lay = o.layout
    o.close()
    assert int(((ro[0]["flags"] & abi.RF_DUP) != 0).sum()) > total // 100, "the input must hold duplicates for --dedup to matter"
    for k, what in enumerate(("read1 results", "read2 results", "pair results")):
        bad = np.nonzero(ro[k] != rg[k])[0]
        assert len(bad) == 0, f"{what} differ at {len(bad)} of {total}, first {bad[:5]}: oracle {ro[k][bad[:3]]} gpu {rg[k][bad[:3]]}"
    bad = np.nonzero(co != cg)[0]
    assert len(bad) == 0, f"{len(bad)} counters differ, first at {bad[:8]}: oracle {co[bad[:8]]} gpu {cg[bad[:8]]}"
    assert int(co[lay.overrep_count[0]:lay.overrep_count[0] + lay.n_overrep[0]].sum()) > 0, "the overrepresentation counters must move"

@pytest.mark.gpu
@pytest.mark.parametrize("paired,L,minlen", [(False, 150, 10), (True, 150, 10), (True, 100, 5), (False, 250, 14)])
def test_gpu_polyg_tails_of_every_length(paired, L, minlen):
    """trimPolyG on G runs from 0 to the whole read (the lane kernel walks 32-base windows of the read's registers)"""
    from test_hostsim_parity import _polyg_tail_reads
    p = abi.default_params(paired, L)
    p.poly_g, p.poly_g_min_len = 1, minlen
    p.adapter_enabled = 0
    if not paired:
        p.adapter_seq_r1 = None
    s1, q1, l1 = _polyg_tail_reads(20000, L, 71)
    args = (s1, q1, l1)
    if paired:
        args += _polyg_tail_reads(20000, L, 72)
    o = oraclelib.Oracle(p)
    g = engines.gpu_engine(p)
    assert g.plan() == "lane"
    ro, rg = o.process(*args), g.process(*args)
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    assert int((l1 - ro[0]["len"] > 40).sum()) > 1000
    for k in range(3):
        if ro[k] is not None:
            bad = np.nonzero(ro[k] != rg[k])[0]
            assert len(bad) == 0, f"result {k} differs at {len(bad)} entries, first {bad[:5]}"
    assert np.array_equal(co, cg)


@pytest.mark.gpu
@pytest.mark.parametrize("paired,L", [(True, 150), (False, 150), (True, 250)])
def test_gpu_plans_agree(paired, L, monkeypatch):
    """the benchmark's option family through each kernel plan (lane + stats / scan + stats / fused): the oracle's
    records and counters from all three"""
    p = abi.default_params(paired, L)
    p.cut_right = 1
    p.poly_g = 1
    if not paired:
        p.adapter_seq_r1 = None
        p.adapter_enabled = 0
    d = synth.synth_pairs(30000, L=L, seed=61, paired=paired, insert_mean=L * 1.4, insert_sd=L * 0.5, polyg_frac=0.1, dup_frac=0.2)
    o = oraclelib.Oracle(p)
    ro, co = o.process(*_args(d, paired)), o.counters()
    o.close()
    for env, want in (({}, "lane"), ({"FASTP_GPU_LANE": "0"}, "split"), ({"FASTP_GPU_LANE": "0", "FASTP_GPU_SPLIT": "0"}, "fused")):
        for k in ("FASTP_GPU_LANE", "FASTP_GPU_SPLIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = engines.gpu_engine(p)
        assert g.plan() == want
        rg, cg = g.process(*_args(d, paired)), g.counters()
        g.close()
        for k in range(3):
            if ro[k] is not None:
                bad = np.nonzero(ro[k] != rg[k])[0]
                assert len(bad) == 0, f"{want}: result {k} differs at {len(bad)} entries, first {bad[:5]}"
        bad = np.nonzero(co != cg)[0]
        assert len(bad) == 0, f"{want}: {len(bad)} counters differ, first at {bad[:8]}"





@pytest.mark.gpu
def test_gpu_stats_cells_at_their_capacity():
    """Every Stats workgroup of a launch with as many units as its 12-bit cell counts hold (4095), all of them identical reads
    with the largest quality character: each workgroup's [count : 12 | quality sum : 20] cells end at 4095 | 93 * 4095.
    Records + every counter against the oracle.  (tests/test_hostsim_parity.py::test_sim_stats_cells_at_their_capacity: one
    workgroup on the emulator)"""
    import test_hostsim_parity as hs
    p = abi.default_params(False, 150)
    p.adapter_seq_r1 = None
    p.adapter_enabled = 0
    p.dup_enabled = 0
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n = 2 * cus * 4095            # two workgroups per CU
    d = hs._saturating_reads(n)
    o = oraclelib.Oracle(p)
    g = engines.gpu_engine(p)
    ro, rg = o.process(d["seq1"], d["qual1"], d["len1"]), g.process(d["seq1"], d["qual1"], d["len1"])
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    assert ro[0].tobytes() == rg[0].tobytes()
    assert np.array_equal(co, cg), int((co != cg).sum())


@pytest.mark.gpu
def test_gpu_stats_joint_table_at_its_capacity():
    """Form 5 of the Stats kernel (fq_stats5.h): every workgroup of a launch with the 16383 units the slab's packed cells hold, all
    of them the same read - the same 16-bit halves of the joint table's cells ('K', its last quality row), then the packed cells of
    what the table has no row for ('~').  (tests/test_hostsim_parity.py::test_sim_stats_joint_table_at_its_capacity: one workgroup)"""
    import test_hostsim_parity as hs
    import torch
    p = abi.default_params(False, 160)
    p.adapter_seq_r1 = None
    p.adapter_enabled = 0
    p.dup_enabled = 0
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for n, ch in ((cus * 16383, "K"), (cus * 16383 + 1, "K"), (cus * 16383, "~")):
        d = hs._filled_reads(n, 160, ch, base="G")
        o = oraclelib.Oracle(p)
        g = engines.gpu_engine(p)
        ro, rg = o.process(d["seq1"], d["qual1"], d["len1"]), g.process(d["seq1"], d["qual1"], d["len1"])
        co, cg = o.counters(), g.counters()
        o.close()
        g.close()
        assert ro[0].tobytes() == rg[0].tobytes()
        assert np.array_equal(co, cg), (n, ch, int((co != cg).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(9))
def test_gpu_stats_every_quality_character(k):
    """every quality character '!' .. '~' around the joint table's edge, N runs, ragged lengths, trimmed ranges (the emulator twin:
    tests/test_hostsim_parity.py::test_sim_stats_every_quality_character, 3000 units): 300 000 units, records + every counter"""
    import test_hostsim_parity as hs
    paired, L, extra = hs.STATS5_RANGE_CASES[k]
    p = abi.default_params(paired, L)
    p.qualified_qual = 48
    p.unqualified_percent_limit = 90
    p.n_base_limit = 50
    p.length_required = 1
    for kk, v in extra.items():
        setattr(p, kk, v)
    d = hs.quality_range_reads(300000, L, 100 + k, paired)
    o = oraclelib.Oracle(p)
    g = engines.gpu_engine(p)
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    ro, rg = o.process(*args), g.process(*args)
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    for i in range(3 if paired else 1):
        assert ro[i].tobytes() == rg[i].tobytes()
    if p.dedup or p.dup_enabled:   # (one launch on the GPU = the oracle's order: the duplicate decisions are the stream's)
        pass
    assert np.array_equal(co, cg), int((co != cg).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(7))
def test_gpu_cut_front_on_the_lane_plan(k):
    """--cut_front on the lane plan (DevParams::front_per_read): a front per read in the lane kernel, in form 5 of the Stats kernel and
    in fq_front_stats_kernel; 200 000 noisy units, records + every counter against the oracle
    (tests/test_hostsim_parity.py::test_sim_cut_front_on_the_lane_plan: 700 units on the emulator)"""
    import test_hostsim_parity as hs
    paired, L, kw = hs.CUT_FRONT_LANE[k]
    p = abi.default_params(paired, L)
    if not paired:
        p.adapter_seq_r1 = None
    p.length_required = 8
    for key, v in kw.items():
        setattr(p, key, v)
    d = synth.noisy_reads(200000, L=L, seed=300 + k, paired=paired)
    o = oraclelib.Oracle(p)
    g = engines.gpu_engine(p)
    assert g.plan() == "lane"
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    ro, rg = o.process(*args), g.process(*args)
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    for i in range(3 if paired else 1):
        bad = np.nonzero(ro[i] != rg[i])[0]
        assert len(bad) == 0, f"case {k}: result {i} differs at {bad[:5]}"
    assert np.array_equal(co, cg), int((co != cg).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [True, False])
def test_gpu_adapter_fasta_on_the_lane_plan(paired):
    """--adapter_fasta in the lane kernel (lists of sequences <= 64 bases): 150 000 units that begin with adapters carrying indels,
    records + adapter events + every counter against the oracle
    (tests/test_hostsim_parity.py::test_sim_adapter_fasta_on_the_lane_plan: 600 units on the emulator)"""
    import test_hostsim_parity as hs
    p = abi.default_params(paired, 150)
    if not paired:
        p.adapter_seq_r1 = None
    abi.set_adapter_fasta(p, hs.FASTA_LANE_LIST)
    d = synth.adapter_indel_reads(150000, L=150, seed=78, paired=paired)
    o = oraclelib.Oracle(p)
    g = engines.gpu_engine(p)
    assert g.plan() == "lane"
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    ro, rg = o.process(*args), g.process(*args)
    co, cg = o.counters(), g.counters()
    eo, eg = o.last_adapter_events, g.last_adapter_events
    o.close()
    g.close()
    for i in range(3 if paired else 1):
        assert ro[i].tobytes() == rg[i].tobytes(), f"records {i} differ"
    assert len(eo) > 1000 and eo.tobytes() == eg.tobytes(), (len(eo), len(eg))
    assert np.array_equal(co, cg), int((co != cg).sum())
