"""The DEVICE code (fastp_amd/csrc) executed by the lock-step SIMT emulator of
tests/hostsim and compared with the CPU oracle - lets the kernels be checked in the
CPU-only container.  The real-GPU parity tests are tests/test_gpu_parity.py."""
import numpy as np
import pytest

import cases
import driver
import engines
import evalport
import gap_util
import golden_util
import oraclelib
import refjson
import synth
from fastp_amd import abi, engine, hostloop

SUPPORTED = list(cases.CASES)


def _both(params, d, paired):
    o = oraclelib.Oracle(params)
    g = engines.sim_engine(params)
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    ro, rg = o.process(*args), g.process(*args)
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    return ro, rg, co, cg


@pytest.mark.parametrize("name", SUPPORTED)
def test_sim_records_and_counters_equal_oracle(name):
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(700, L=150, seed=21, paired=paired, **skw)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    ro, rg, co, cg = _both(params, d, paired)
    for k, what in enumerate(("r1", "r2", "pair")):
        if ro[k] is not None:
            bad = np.nonzero(ro[k] != rg[k])[0]
            assert len(bad) == 0, f"{name}: {what} differs at {bad[:5]}: oracle {ro[k][bad[:3]]} device {rg[k][bad[:3]]}"
    so = np.sort(ro[3], order=["read", "pos"])
    sg = np.sort(rg[3], order=["read", "pos"])
    assert np.array_equal(so, sg), f"{name}: correction lists differ"
    assert np.array_equal(co, cg), f"{name}: {int((co != cg).sum())} counters differ"


@pytest.mark.parametrize("name,table", [("pe_noadapter_dedup", 0), ("pe_default", 0), ("se_default_noadapter", 0), ("pe_noadapter_dedup", 1)])
def test_sim_duplicates_when_workgroups_run_out_of_order(name, table, monkeypatch):
    """Duplicate's decision must not depend on which unit reaches a bloom bit first: the emulator runs the workgroups last
    to first, so the unit that wins a bit is usually NOT the first in input order (the winners / finish kernels have to
    flip it); many planted duplicates, several launches.  table=1: the first (probe / resolve) form of the kernels."""
    monkeypatch.setenv("FASTP_SIM_REVERSE_BLOCKS", "1")
    monkeypatch.setenv("FASTP_GPU_MAX_TILES_PER_BLOCK", "2")
    if table:
        monkeypatch.setenv("FASTP_GPU_DUP_TABLE", "1")
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(2400, L=150, seed=31, paired=paired, **skw)
    rng = np.random.default_rng(32)     # exact copies of earlier units (sequencing noise makes the generator's own rare)
    dst = rng.choice(np.arange(1, 2400), size=800, replace=False)
    src = (rng.random(800) * dst).astype(np.int64)
    for k in d:
        d[k][dst] = d[k][src]
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    ro, rg, co, cg = _both(params, d, paired)
    dups = int((ro[0]["flags"] & abi.RF_DUP != 0).sum())
    assert dups > 100, dups
    for k in range(3):
        if ro[k] is not None:
            assert np.array_equal(ro[k], rg[k]), f"{name}: records {k} differ"
    assert np.array_equal(co, cg)


@pytest.mark.parametrize("switch", ["FASTP_GPU_CLAIM_FUSED", "FASTP_GPU_DEDUP_FOLD"])
def test_sim_dedup_without_the_fused_claim(switch, monkeypatch):
    """--dedup on the lane plan with the claim step taken out of the lane kernel (the A/B switches of DESIGN.md 5): the hash
    pre-pass has to decide then - with FASTP_GPU_CLAIM_FUSED=0 alone Duplicate used not to run at all (round 5's advisor finding)"""
    monkeypatch.setenv(switch, "0")
    paired, flags, pf, skw = cases.CASES["pe_noadapter_dedup"]
    d = synth.synth_pairs(1500, L=150, seed=41, paired=paired, **skw)
    rng = np.random.default_rng(42)
    dst = rng.choice(np.arange(1, 1500), size=500, replace=False)
    src = (rng.random(500) * dst).astype(np.int64)
    for k in d:
        d[k][dst] = d[k][src]
    params = cases.finalize_params("pe_noadapter_dedup", pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    ro, rg, co, cg = _both(params, d, paired)
    assert int((ro[0]["flags"] & abi.RF_DUP != 0).sum()) > 100
    for k in range(3):
        assert np.array_equal(ro[k], rg[k]), f"records {k} differ"
    assert np.array_equal(co, cg)


@pytest.mark.parametrize("name,log2", [("pe_default", 1), ("pe_default", 2), ("se_default_noadapter", 1), ("pe_correction", 1), ("pe_noadapter_dedup", 2)])
def test_sim_lane_chunk_pool(name, log2, monkeypatch):
    """the lane kernel's chunks behind the workgroups' shares (LaneArgs::pool: taken from a global counter that only counts up, its
    base moved on by the host per launch): half / a quarter of a launch's chunks in the pool, three launches on one context - every
    unit exactly once (records, counters), whatever the counter's state between launches"""
    monkeypatch.setenv("FASTP_GPU_LANE_POOL_LOG2", str(log2))
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(2600, L=150, seed=61, paired=paired, **skw)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    o = oraclelib.Oracle(params)
    g = engines.sim_engine(params)
    assert g.plan() == "lane"
    for lo, hi in ((0, 1100), (1100, 1200), (1200, 2600)):
        args = tuple(d[k][lo:hi] for k in (("seq1", "qual1", "len1", "seq2", "qual2", "len2") if paired else ("seq1", "qual1", "len1")))
        ro, rg = o.process(*args), g.process(*args)
        for k in range(3):
            if ro[k] is not None:
                assert np.array_equal(ro[k], rg[k]), f"{name}: records {k} of units [{lo}, {hi}) differ"
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    assert np.array_equal(co, cg), f"{name}: {int((co != cg).sum())} counters differ"


def test_sim_overrep_with_correction_reads_the_engines_own_list():
    """-p with -c on the lane plan: the POST overrepresentation counts are taken from the launch's own correction list (sized for
    an edit at every base: it cannot overflow).  A caller's list that is too small for the batch's edits - the results then hold
    its first entries only and the submit returns FASTP_GPU_E_CAPACITY - must not change a counter (round 5's advisor finding: the
    analysis used to read the caller's)"""
    name = "pe_overrep_correction"
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(1500, L=150, seed=21, paired=paired, **skw)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    o = oraclelib.Oracle(params)
    g = engines.sim_engine(params)
    assert g.plan() == "lane"
    g.corr_capacity = 2
    args = (d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    ro = o.process(*args)
    assert len(ro[3]) > 50, len(ro[3])
    with pytest.raises(engine.EngineError, match="correction list capacity exceeded"):   # the caller is told that its list is short
        g.process(*args)
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    assert np.array_equal(co, cg), f"{int((co != cg).sum())} counters differ"


@pytest.mark.parametrize("level,L,paired", [(1, 150, True), (3, 150, True), (1, 37, True), (1, 250, True), (1, 150, False), (-3, 100, True)])
def test_sim_duplicate_hash_bit_positions_equal_oracle(level, L, paired, monkeypatch):
    """the hash itself (Duplicate::seq2intvector mod mBufLenInBits), not only the decisions it leads to: every
    length incl. ragged read-1 lengths (read 2 continues the position index there), N bases, every buffer"""
    import torch
    import shard_util
    if level < 0:   # the multiply form of the hash (the path of accuracy level 6, whose 32 GiB of bitmaps the emulator cannot afford)
        monkeypatch.setenv("FASTP_GPU_HASH_GENERIC", "1")
        level = -level
    p = abi.default_params(paired, L)
    p.dup_accuracy_level = level
    if not paired:
        p.adapter_seq_r1 = None
    d = synth.synth_pairs(500, L=L, seed=3 + level, paired=paired, ragged_frac=0.6, insert_mean=L * 1.2, insert_sd=L * 0.4)
    g = engines.sim_engine(p)
    got = shard_util.device_bit_positions(g, d, torch.device("cpu"))
    g.close()
    want = shard_util.oracle_bit_positions(level, d)
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} bit positions differ"


@pytest.mark.parametrize("name", cases.GAP_CASES)
def test_sim_one_gap_accept_paths(name):
    """the oracle must ACCEPT one-gap overlaps / one-gap adapter matches on these inputs (counted), then the
    device code's closed-form versions are compared on them"""
    paired, flags, pf, skw = cases.CASES[name]
    n = 3000
    d = synth.synth_pairs(n, L=150, seed=31, paired=paired, **skw)
    params = pf(150)
    ro, rg, co, cg = _both(params, d, paired)
    if "allow_gap" in name:
        tot, neg = gap_util.gap_overlap_pairs(d, params)
        assert tot > 100 and neg > 40, f"{name}: only {tot} one-gap overlaps ({neg} with a negative offset)"
        assert gap_util.gap_trimmed_pairs(ro[0], ro[2]) > 40
    else:
        assert gap_util.gap_adapter_trims(d["seq1"], d["len1"], ro[0], bytes(params.adapter_seq_r1)) > 500
        if paired:
            assert gap_util.gap_adapter_trims(d["seq2"], d["len2"], ro[1], bytes(params.adapter_seq_r2)) > 500
    for k, what in enumerate(("r1", "r2", "pair")):
        if ro[k] is not None:
            bad = np.nonzero(ro[k] != rg[k])[0]
            assert len(bad) == 0, f"{name}: {what} differs at {bad[:5]}: oracle {ro[k][bad[:3]]} device {rg[k][bad[:3]]}"
    assert np.array_equal(np.sort(ro[3], order=["read", "pos"]), np.sort(rg[3], order=["read", "pos"]))
    assert np.array_equal(co, cg), f"{name}: {int((co != cg).sum())} counters differ"


@pytest.mark.parametrize("name", ["pe_default", "pe_correction", "se_adapter_cut", "testdata_pe", "pe_overlapped_out_trims",
                                  "pe_merge_overlapped_out", "pe_merge_overlapped_out_trims",
                                  "pe_exotic_default", "pe_exotic_merge", "pe_exotic_dedup_adapters", "se_exotic_adapter"])
def test_sim_matches_reference_golden(name):
    fq1, fq2, meta = golden_util.load(name)
    params = golden_util.params_for(name, max_len=152, fq1=fq1, fq2=fq2)
    eng = engines.sim_engine(params)
    outs, ctr, rep = driver.run_engine(eng, params, fq1, fq2, umi=golden_util.umi_for(name))
    eng.close()
    golden_util.check_against_golden(name, outs, rep, meta)


def test_sim_other_read_lengths_and_tile_shapes(monkeypatch):
    """2x250 and short reads; a non-default tile / workgroup shape must not change anything"""
    for L, tile, threads in ((250, 0, 512), (75, 12, 128), (100, 40, 256)):
        if tile:
            monkeypatch.setenv("FASTP_GPU_TILE", str(tile))
        monkeypatch.setenv("FASTP_GPU_THREADS", str(threads))
        p = abi.default_params(True, L)
        p.cut_right = 1
        p.correction = 1
        p.poly_g = 1
        d = synth.synth_pairs(300, L=L, seed=5, insert_mean=L * 1.3, insert_sd=L * 0.4, polyg_frac=0.1)
        ro, rg, co, cg = _both(p, d, True)
        for k in range(3):
            assert ro[k].tobytes() == rg[k].tobytes(), (L, k)
        assert np.array_equal(co, cg)


@pytest.mark.parametrize("paired,L", [(True, 150), (False, 150), (True, 100)])
def test_sim_three_kernel_plans_agree(paired, L, monkeypatch):
    """the benchmark's option family through each plan (lane + stats kernels / scan + stats kernels / fused kernel):
    the same records and counters as the oracle from all three"""
    p = abi.default_params(paired, L)
    p.cut_right = 1
    p.poly_g = 1
    if not paired:
        p.adapter_seq_r1 = None
        p.adapter_enabled = 0
    d = synth.synth_pairs(900, L=L, seed=61, paired=paired, insert_mean=L * 1.4, insert_sd=L * 0.5, polyg_frac=0.1, dup_frac=0.2)
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    o = oraclelib.Oracle(p)
    ro, co = o.process(*args), o.counters()
    o.close()
    for env, want in (({}, "lane"), ({"FASTP_GPU_LANE": "0"}, "split"), ({"FASTP_GPU_LANE": "0", "FASTP_GPU_SPLIT": "0"}, "fused")):
        for k in ("FASTP_GPU_LANE", "FASTP_GPU_SPLIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = engines.sim_engine(p)
        assert g.plan() == want
        rg, cg = g.process(*args), g.counters()
        g.close()
        for k in range(3):
            if ro[k] is not None:
                assert ro[k].tobytes() == rg[k].tobytes(), (want, k)
        assert np.array_equal(co, cg), (want, int((co != cg).sum()))


def _polyg_tail_reads(n, L, seed):
    """reads that end in G runs of 0 .. L bases with a few other bases sprinkled in (PolyX::trimPolyG's mismatch rules),
    some entirely G, lengths ragged"""
    rng = np.random.default_rng(seed)
    stride = (L + 7) // 8 * 8
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, stride))]
    run = np.minimum(rng.integers(0, L + 30, size=n), L)
    run = np.where(rng.random(n) < 0.3, rng.integers(0, 40, size=n), run)
    lens = np.where(rng.random(n) < 0.3, rng.integers(1, L + 1, size=n), L).astype(np.int32)
    j = np.arange(stride)[None, :]
    in_run = (j >= (lens - run)[:, None]) & (j < lens[:, None])
    noise = rng.random((n, stride)) < rng.choice([0.0, 0.05, 0.12, 0.3], size=(n, 1))
    seq = np.where(in_run & ~noise, ord("G"), seq).astype(np.uint8)
    seq = np.where(rng.random((n, stride)) < 0.01, ord("N"), seq).astype(np.uint8)
    qual = rng.integers(35, 41, size=(n, stride)).astype(np.uint8) + 33
    seq[j >= lens[:, None]] = 0
    qual[j >= lens[:, None]] = 0
    return seq, qual, lens


@pytest.mark.parametrize("paired,L,minlen", [(False, 150, 10), (True, 150, 10), (True, 100, 5), (False, 250, 14)])
def test_sim_polyg_tails_of_every_length(paired, L, minlen):
    """trimPolyG on G runs from 0 to the whole read, with mismatches inside the run: the lane kernel walks 32-base
    windows of the read's registers; runs longer than a window take the outer loop"""
    p = abi.default_params(paired, L)
    p.poly_g, p.poly_g_min_len = 1, minlen
    p.adapter_enabled = 0
    if not paired:
        p.adapter_seq_r1 = None
    s1, q1, l1 = _polyg_tail_reads(1500, L, 71)
    args = (s1, q1, l1)
    if paired:
        args += _polyg_tail_reads(1500, L, 72)
    o = oraclelib.Oracle(p)
    g = engines.sim_engine(p)
    assert g.plan() == "lane"
    ro, rg = o.process(*args), g.process(*args)
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    assert int((ro[0]["len"] < l1).sum()) > 400, "the inputs must make trimPolyG cut"
    assert int((l1 - ro[0]["len"] > 40).sum()) > 100, "runs longer than one window must occur"
    for k in range(3):
        if ro[k] is not None:
            bad = np.nonzero(ro[k] != rg[k])[0]
            assert len(bad) == 0, f"result {k} differs at {bad[:5]}: oracle {ro[k][bad[:3]]} device {rg[k][bad[:3]]}"
    assert np.array_equal(co, cg)


def test_sim_out_of_scope_parameters_fail_loudly():
    """the engine never falls back: parameters outside the device path are errors"""
    for field, value, code in (("max_len", 513, abi.E_TOO_LONG), ("insert_size_max", 5000, abi.E_INVALID),
                               ("abi_version", 99, abi.E_INVALID), ("unqualified_percent_limit", -1, abi.E_INVALID),
                               ("adapter_seq_r1", b"ACGTN", abi.E_INVALID), ("adapter_seq_r1", b"A" * (abi.MAX_ADAPTER_LEN + 1), abi.E_UNSUPPORTED)):
        p = abi.default_params(True, 150)
        setattr(p, field, value)
        with pytest.raises(engine.EngineError) as e:
            engines.sim_engine(p)
        assert e.value.code == code, field


@pytest.mark.parametrize("k", range(len(cases.TRIM_STRESS)))
def test_sim_trim_and_cut_stress(k):
    """Filter::trimAndCut via the predicate-mask bit scans vs the oracle's literal loops"""
    paired, kw = cases.TRIM_STRESS[k]
    p = abi.default_params(paired, 150)
    if not paired:
        p.adapter_seq_r1 = None
    for key, v in kw.items():
        setattr(p, key, v)
    d = synth.noisy_reads(500, L=150, seed=100 + k, paired=paired)
    ro, rg, co, cg = _both(p, d, paired)
    for i in range(3 if paired else 1):
        bad = np.nonzero(ro[i] != rg[i])[0]
        assert len(bad) == 0, f"stress {k}: result {i} differs at {bad[:5]}: oracle {ro[i][bad[:3]]} device {rg[i][bad[:3]]}"
    assert np.array_equal(co, cg)


@pytest.mark.parametrize("k", range(len(cases.OVERLAP_STRESS)))
def test_sim_overlap_stress(k):
    """OverlapAnalysis::analyze (prefilter + exact verify + scan-order key) vs the oracle's literal scan"""
    p = abi.default_params(True, 150)
    for key, v in cases.OVERLAP_STRESS[k].items():
        setattr(p, key, v)
    d = synth.overlap_pairs(600, L=150, seed=300 + k)
    ro, rg, co, cg = _both(p, d, True)
    for i in range(3):
        bad = np.nonzero(ro[i] != rg[i])[0]
        assert len(bad) == 0, f"overlap stress {k}: result {i} differs at {bad[:5]}: oracle {ro[i][bad[:3]]} device {rg[i][bad[:3]]}"
    assert np.array_equal(np.sort(ro[3], order=["read", "pos"]), np.sort(rg[3], order=["read", "pos"]))
    assert np.array_equal(co, cg)


@pytest.mark.parametrize("k", range(len(cases.MERGE_STRESS)))
def test_sim_merge_stress(k):
    """merge mode: second overlap analysis, OverlapAnalysis::merge, passFilter(merged), stats of the merged read"""
    p = abi.default_params(True, 150)
    for key, v in cases.MERGE_STRESS[k].items():
        setattr(p, key, v)
    d = synth.overlap_pairs(600, L=150, seed=700 + k, err=0.01, n_rate=0.01)
    ro, rg, co, cg = _both(p, d, True)
    for i in range(3):
        bad = np.nonzero(ro[i] != rg[i])[0]
        assert len(bad) == 0, f"merge stress {k}: result {i} differs at {bad[:5]}: oracle {ro[i][bad[:3]]} device {rg[i][bad[:3]]}"
    assert np.array_equal(np.sort(ro[3], order=["read", "pos"]), np.sort(rg[3], order=["read", "pos"]))
    bad = np.nonzero(co != cg)[0]
    assert len(bad) == 0, f"merge stress {k}: counters differ at {bad[:8]}: oracle {co[bad[:8]]} device {cg[bad[:8]]}"


def check_merge_lane(mk, k, plan="lane"):
    p, sets = cases.merge_lane_case(k)
    for d in sets:
        args = (d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
        o = oraclelib.Oracle(p)
        g = mk(p)
        assert g.plan() == plan
        ro, rg = o.process(*args), g.process(*args)
        co, cg = o.counters(), g.counters()
        o.close()
        g.close()
        for i in range(3):
            bad = np.nonzero(ro[i] != rg[i])[0]
            assert len(bad) == 0, f"merge on the lane plan {k}: result {i} differs at {bad[:5]}: oracle {ro[i][bad[:3]]} device {rg[i][bad[:3]]}"
        assert np.array_equal(np.sort(ro[3], order=["read", "pos"]), np.sort(rg[3], order=["read", "pos"]))
        bad = np.nonzero(co != cg)[0]
        assert len(bad) == 0, f"merge on the lane plan {k}: counters differ at {bad[:8]}: oracle {co[bad[:8]]} device {cg[bad[:8]]}"
        assert int(co[g.layout.merged_pairs]) > 50   # (the case does merge)


@pytest.mark.parametrize("k", range(len(cases.MERGE_LANE)))
@pytest.mark.parametrize("slow", [0, 1])
def test_sim_merge_on_the_lane_plan(k, slow, monkeypatch):
    """the second overlap analysis, the merged read's filter, its Stats (read 1's part, the reverse-complemented part of read 2 at
    the merged read's cycles, the junction 5-mers), --include_unmerged, -c's edits in either part - every record and counter
    against the oracle.  slow: every merged read's second part counted by the lane kernel itself (the path a pair takes whose
    second part holds an edited base)"""
    if slow:
        monkeypatch.setenv("FASTP_GPU_TEST_MERGE_SLOW", "1")
    check_merge_lane(engines.sim_engine, k)


@pytest.mark.parametrize("k", [0, 1, 3])
def test_sim_merge_lane_and_fused_plans_agree(k, monkeypatch):
    """the same merge-mode input through the lane plan and (FASTP_GPU_LANE=0) through the fused kernel the options took until
    round 5: the same records, corrections and counters from both, equal to the oracle's"""
    p, sets = cases.merge_lane_case(k, n=500)
    d = sets[1]
    args = (d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    o = oraclelib.Oracle(p)
    ro, co = o.process(*args), o.counters()
    o.close()
    for env, want in (({}, "lane"), ({"FASTP_GPU_LANE": "0"}, "fused")):
        monkeypatch.delenv("FASTP_GPU_LANE", raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        g = engines.sim_engine(p)
        assert g.plan() == want
        rg, cg = g.process(*args), g.counters()
        g.close()
        for i in range(3):
            assert ro[i].tobytes() == rg[i].tobytes(), (want, i)
        assert np.array_equal(np.sort(ro[3], order=["read", "pos"]), np.sort(rg[3], order=["read", "pos"])), want
        assert np.array_equal(co, cg), (want, int((co != cg).sum()))


def test_sim_merge_on_the_lane_plan_rows_not_16_byte_aligned():
    """a device batch whose arrays start 4 bytes off a 16-byte boundary: merge mode has no tile form of this plan, the engine moves
    the rows to aligned arrays of its own - the same records and counters"""
    p, sets = cases.merge_lane_case(1, n=400)
    d = sets[1]
    g = engines.sim_engine(p)
    n = len(d["len1"])
    packed = [engine.pack_ascii(g.lib, p.max_len, d["seq" + m], d["qual" + m], d["len" + m]) for m in "12"]
    want = g.submit_packed(*packed[0], *packed[1])
    cw = g.counters()
    g.close()
    g = engines.sim_engine(p)
    keep = []

    def off4(a):   # a copy of the array that starts 4 bytes behind a 16-byte boundary
        raw = np.zeros(a.nbytes + 64, dtype=np.uint8)
        start = (-raw.ctypes.data) % 16 + 4
        raw[start:start + a.nbytes] = a.view(np.uint8).reshape(-1)
        keep.append(raw)
        return raw.ctypes.data + start
    b = abi.Batch()
    b.n, b.flags = n, abi.BATCH_STAT_ISIZE
    b.seq1, b.qual1, b.len1 = off4(packed[0][0]), off4(packed[0][1]), packed[0][2].ctypes.data
    b.seq2, b.qual2, b.len2 = off4(packed[1][0]), off4(packed[1][1]), packed[1][2].ctypes.data
    assert b.seq1 % 16 == 4 and b.qual2 % 16 == 4
    r1 = np.zeros(n, dtype=abi.READ_RESULT_DTYPE)
    r2 = np.zeros(n, dtype=abi.READ_RESULT_DTYPE)
    pr = np.zeros(n, dtype=abi.PAIR_RESULT_DTYPE)
    corr = np.zeros(n * 32, dtype=abi.CORRECTION_DTYPE)
    nc = np.zeros(1, dtype=np.int32)
    res = abi.Results()
    res.r1, res.r2, res.pair = r1.ctypes.data, r2.ctypes.data, pr.ctypes.data
    res.corrections, res.corrections_capacity, res.n_corrections = corr.ctypes.data, len(corr), nc.ctypes.data
    g.submit_device(b, res)
    g.synchronize()
    cg = g.counters()
    g.close()
    assert r1.tobytes() == want[0].tobytes() and r2.tobytes() == want[1].tobytes() and pr.tobytes() == want[2].tobytes()
    assert np.array_equal(np.sort(corr[:nc[0]], order=["read", "pos"]), np.sort(want[3], order=["read", "pos"]))
    assert np.array_equal(cw, cg)


def test_sim_stats_work_list_overflow():
    """one-pass Stats: so many N-containing quality dwords that the LDS work list overflows and the
    fast path has to run the general code in place"""
    p = abi.default_params(True, 150)
    p.cut_right = 1
    p.n_base_limit = 100
    d = synth.noisy_reads(400, L=150, seed=77, paired=True, n_rate=0.35)
    ro, rg, co, cg = _both(p, d, True)
    for i in range(3):
        assert ro[i].tobytes() == rg[i].tobytes()
    assert np.array_equal(co, cg)


def test_sim_overrep_streamed_and_split_launches(monkeypatch):
    """overrepresentation sampling is a position in the run's read stream: feeding the reads in
    several batches, each split into several launches, must give the oracle's one-stream counts"""
    monkeypatch.setenv("FASTP_GPU_TILE", "32")
    monkeypatch.setenv("FASTP_GPU_MAX_TILES_PER_BLOCK", "2")   # 3 "CUs" x 2 tiles x 32 pairs per launch
    name = "pe_overrep"
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(900, L=150, seed=5, paired=True, **skw)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d["seq2"], d["len2"])
    assert params.n_overrep_seqs1 > 0
    o = oraclelib.Oracle(params)
    g = engines.sim_engine(params)
    o.process(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    for a, b in ((0, 250), (250, 251), (251, 900)):
        sl = {k: v[a:b] for k, v in d.items()}
        g.process(sl["seq1"], sl["qual1"], sl["len1"], sl["seq2"], sl["qual2"], sl["len2"])
    co, cg = o.counters(), g.counters()
    o.close()
    g.close()
    lay = o.layout
    assert co[lay.overrep_count[0]: lay.overrep_count[0] + lay.n_overrep[0]].sum() > 0
    bad = np.nonzero(co != cg)[0]
    assert len(bad) == 0, (bad[:10], co[bad[:10]], cg[bad[:10]])


def _fastq_text(n=300, L=150, seed=9, eol=b"\n", trailing=True):
    d = synth.synth_pairs(n, L=L, seed=seed, paired=False)
    txt = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    txt = txt.replace(b"\n", eol)
    if not trailing:
        txt = txt[:-len(eol)]
    return txt


@pytest.mark.parametrize("eol,trailing", [(b"\n", True), (b"\r\n", True), (b"\r", True), (b"\n", False), (b"\r\n", False)])
def test_sim_device_fastq_parse_equals_reader_and_packer(eol, trailing):
    """FASTQ text -> packed rows on the device == FastqReader's line splitting + the host packer"""
    import parse_util
    g = engines.sim_engine(abi.default_params(False, 150))
    txt = _fastq_text(eol=eol, trailing=trailing)
    exp = parse_util.expected(txt, 150, None, True)
    info, seq, qual, lens, loff, llen = parse_util.run_numpy(g, txt, 150, 1000, True)
    assert info.n_records == 300 and info.first_bad == -1 and info.consumed == len(txt)
    assert np.array_equal(seq, exp[0]) and np.array_equal(qual, exp[1]) and np.array_equal(lens, exp[2])
    assert np.array_equal(loff, exp[3]) and np.array_equal(llen, exp[4])
    # the packed rows equal what the C packer makes from the same reads
    b = hostloop.parse_fastq(txt.replace(eol, b"\n") + (b"" if trailing else b"\n"))
    s2, q2, l2 = engine.pack_ascii(g.lib, 150, b.seq, b.qual, b.lens)
    assert np.array_equal(seq, s2) and np.array_equal(qual, q2) and np.array_equal(lens, l2)
    g.close()


def test_sim_device_fastq_parse_chunks_limits_and_errors():
    import parse_util
    g = engines.sim_engine(abi.default_params(False, 150))
    txt = _fastq_text(n=120, eol=b"\r\n")
    # a chunk cut in the middle of a record (and of a \r\n): only complete records, consumed tells where to resume
    for cut in (len(txt) // 2, len(txt) // 2 + 1, txt.index(b"\r\n", 5000) + 1, 17, 0):
        chunk = txt[:cut]
        exp = parse_util.expected(chunk, 150, None, False)
        info, seq, qual, lens, loff, llen = parse_util.run_numpy(g, chunk, 150, 1000, False)
        assert info.n_records == len(exp[2]) and info.consumed == exp[5], (cut, info.n_records, info.consumed, exp[5])
        assert np.array_equal(seq, exp[0]) and np.array_equal(qual, exp[1]) and np.array_equal(lens, exp[2])
        rest = txt[info.consumed:]
        info2, *_ = parse_util.run_numpy(g, rest, 150, 1000, True)
        assert info.n_records + info2.n_records == 120
    # max_records caps the batch
    info, seq, qual, lens, loff, llen = parse_util.run_numpy(g, txt, 150, 7, True)
    exp = parse_util.expected(txt, 150, 7, True)
    assert info.n_records == 7 and info.consumed == exp[5] and np.array_equal(qual, exp[1])
    # malformed chunks are refused, never repaired on the device
    lines = txt.split(b"\r\n")
    for mutate in ("name", "plus", "length", "alphabet", "toolong", "qual_low", "qual_del"):
        ls = list(lines)
        if mutate == "name":
            ls[4 * 5] = b"X" + ls[4 * 5][1:]
        elif mutate == "plus":
            ls[4 * 5 + 2] = b"-"
        elif mutate == "length":
            ls[4 * 5 + 3] = ls[4 * 5 + 3][:-1]
        elif mutate == "alphabet":
            ls[4 * 5 + 1] = b"R" + ls[4 * 5 + 1][1:]
        elif mutate == "qual_low":    # below '!'
            ls[4 * 5 + 3] = ls[4 * 5 + 3][:6] + b" " + ls[4 * 5 + 3][7:]
        elif mutate == "qual_del":    # above '~'
            ls[4 * 5 + 3] = ls[4 * 5 + 3][:9] + b"\x7f" + ls[4 * 5 + 3][10:]
        else:
            ls[4 * 5 + 1] = ls[4 * 5 + 1] + b"ACGT" * 10
            ls[4 * 5 + 3] = ls[4 * 5 + 3] + b"IIII" * 10
        info, *_ = parse_util.run_numpy(g, b"\r\n".join(ls), 150, 1000, True, check=False)
        if mutate == "alphabet":   # not an error: the record is listed for the text kernel (fastp_gpu_parse_exotic)
            assert info.rc == 0 and info.first_bad == -1 and info.n_exotic == 1 and list(g.parse_exotic()) == [5]
            continue
        assert info.rc == abi.E_INVALID and info.first_bad == 5, (mutate, info.rc, info.first_bad)
        assert info.n_exotic == 0
    g.close()


FORMAT_CASES = ["pe_default", "pe_cut_front_tail", "pe_adapter_seq", "pe_correction", "pe_filters", "pe_noadapter_dedup",
                "pe_allow_gap", "pe_adapter_fasta", "se_default_noadapter", "se_adapter_cut", "se_polyx_complexity"]


def _format_case(mk_engine, mem, name, n, eol=b"\n"):
    import format_util
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(n, L=150, seed=77, paired=paired, **skw)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    fq2 = synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2) if paired else None
    ref = mk_engine(params)
    want, _, _ = driver.run_engine(ref, params, fq1, fq2, pack=n, stride=abi.qual_stride(150))
    ref.close()
    g = mk_engine(params)
    rc, o1, o2, lens = format_util.run(g, mem, params, fq1.replace(b"\n", eol), fq2.replace(b"\n", eol) if paired else None, 150)
    g.close()
    assert rc == 0
    assert o1 == bytes(want.out1), f"{name}: out1 differs"
    if paired:
        assert o2 == bytes(want.out2), f"{name}: out2 differs"
    assert lens[0] == len(want.out1)
    return want


@pytest.mark.parametrize("name", FORMAT_CASES)
def test_sim_device_fastq_format_equals_host_writer(name):
    """results + parsed text -> out1/out2 text on the device == hostloop.apply_results' out1/out2"""
    import format_util
    want = _format_case(engines.sim_engine, format_util.NumpyMem(), name, 600)
    assert len(want.out1) > 0


def test_sim_device_fastq_format_crlf_limits_and_errors():
    import format_util
    _format_case(engines.sim_engine, format_util.NumpyMem(), "pe_correction", 300, eol=b"\r\n")
    # too small an output buffer: E_OVERFLOW, needed sizes reported, nothing written past the capacity
    paired, flags, pf, skw = cases.CASES["se_default_noadapter"]
    d = synth.synth_pairs(100, L=150, seed=5, paired=False)
    fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    p = pf(150)
    g = engines.sim_engine(p)
    rc, o1, _, lens = format_util.run(g, format_util.NumpyMem(), p, fq1, None, 150, out_slack=-(len(fq1) // 2))
    assert rc == abi.E_OVERFLOW and lens[0] > len(o1)
    g.close()
    for name in ("pe_merge", "pe_umi_per_read"):
        paired, flags, pf, skw = cases.CASES[name]
        d = synth.synth_pairs(50, L=150, seed=5)
        p = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d["seq2"], d["len2"])
        g = engines.sim_engine(p)
        rc, *_ = format_util.run(g, format_util.NumpyMem(), p, synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1),
                                 synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2), 150)
        assert rc == abi.E_UNSUPPORTED
        g.close()


STREAM_CASES = [("pe_filters", True, True), ("pe_filters", True, False), ("pe_filters", False, True), ("pe_merge", True, False),
                ("pe_merge_unmerged", True, False), ("pe_umi_per_read", True, True), ("se_umi_read1", True, False),
                ("pe_correction", True, True), ("se_adapter_cut", True, False), ("pe_noadapter_dedup", True, True),
                ("se_polyx_complexity", True, False), ("pe_default", False, False)]


def _streams_case(mk_engine, mem, name, n, want_failed, want_unpaired, umi_extra=(), eol=b"\n", seed=77):
    """fastp_gpu_format_streams' six streams == hostloop.apply_results' (the Python restatement of the worker loop's
    string side, itself pinned to fastp_ref's files by the golden tests)"""
    import format_util
    from fastp_amd import hostloop
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(n, L=150, seed=seed, paired=paired, **skw)
    params = cases.finalize_params(name, pf(150), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    fq2 = synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2) if paired else None
    umi = cases.UMI.get(name)
    if umi is not None:
        umi = tuple(umi) + tuple(umi_extra)
    editor = hostloop.UmiNameEditor(umi[0], umi[1], *[x for x in umi[2:]]) if umi else None
    ref = mk_engine(params)
    want, _, _ = driver.run_engine(ref, params, fq1, fq2, pack=n, stride=abi.qual_stride(150), want_failed=want_failed,
                                   want_unpaired=want_unpaired, umi=editor)
    ref.close()
    g = mk_engine(params)
    rc, got, lens = format_util.run_streams(g, mem, params, fq1.replace(b"\n", eol), fq2.replace(b"\n", eol) if paired else None,
                                            150, want_failed, want_unpaired, umi)
    g.close()
    assert rc == 0
    for k in format_util.STREAMS:
        w = getattr(want, k, None)
        w = bytes(w) if w is not None else b""
        assert got[k] == w, f"{name}: stream {k} differs ({len(got[k])} vs {len(w)} bytes)"
    return got


@pytest.mark.parametrize("name,want_failed,want_unpaired", STREAM_CASES)
def test_sim_device_all_streams_equal_host_writer(name, want_failed, want_unpaired):
    import format_util
    got = _streams_case(engines.sim_engine, format_util.NumpyMem(), name, 500, want_failed, want_unpaired)
    assert sum(len(v) for v in got.values()) > 0
    if "merge" in name:
        assert len(got["merged"]) > 0 and b" merged_" in got["merged"]
    if want_failed and name in ("pe_filters", "se_polyx_complexity"):
        assert b" failed_" in got["failed"]
    if want_unpaired and name == "pe_filters":
        assert len(got["unpaired1"]) > 0 and len(got["unpaired2"]) > 0


def test_sim_device_all_streams_umi_prefix_crlf_and_overflow():
    import format_util
    got = _streams_case(engines.sim_engine, format_util.NumpyMem(), "pe_umi_per_read", 200, True, False, umi_extra=(b"UMI", b"-"))
    assert b"-UMI_" in got["out1"]
    _streams_case(engines.sim_engine, format_util.NumpyMem(), "pe_merge", 200, True, False, eol=b"\r\n")
    paired, flags, pf, skw = cases.CASES["pe_filters"]
    d = synth.synth_pairs(200, L=150, seed=5)
    p = cases.finalize_params("pe_filters", pf(150), d["seq1"], d["len1"], d["seq2"], d["len2"])
    g = engines.sim_engine(p)
    rc, got, lens = format_util.run_streams(g, format_util.NumpyMem(), p, synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1),
                                            synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2), 150, shrink=1)
    assert rc == abi.E_OVERFLOW and lens[1] > 100 and lens[0] == len(got["out1"])
    g.close()


def _odd_records(mate, rng, n=160):
    """hand-made records: names with no / one / several spaces, strand lines that repeat the name, reads of length 0..40
    (shorter than the UMI, shorter than the filters), N runs"""
    out = []
    for i in range(n):
        L = int(rng.integers(0, 41)) if i % 3 else int(rng.integers(0, 4))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
        qual = bytes(rng.integers(35, 75, size=L, dtype=np.uint8))
        name = [b"@r%d" % i, b"@r%d %d:N:0:ACGT" % (i, mate), b"@r%d  two  spaces %d" % (i, mate), b"@%d/%d" % (i, mate)][i % 4]
        strand = b"+" if i % 5 else b"+" + name[1:]
        out.append(name + b"\n" + seq + b"\n" + strand + b"\n" + qual + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("paired,umi", [(True, ("per_read", 6)), (True, ("read2", 3)), (False, ("read1", 5)), (True, None), (False, None)])
def test_sim_device_all_streams_on_odd_records(paired, umi):
    """the stream formatter on records a generator does not make: empty reads, names without a space, strand lines that
    carry the name, UMIs longer than the read - against the host writer (hostloop.apply_results)"""
    import format_util
    from fastp_amd import hostloop
    rng = np.random.default_rng(17)
    fq1 = _odd_records(1, rng)
    fq2 = _odd_records(2, rng) if paired else None
    p = abi.default_params(paired, 48)
    p.length_required = 8
    p.cut_right = 1
    if umi is not None:
        if umi[0] in ("read1", "per_read"):
            p.umi_len1 = umi[1]
        if paired and umi[0] in ("read2", "per_read"):
            p.umi_len2 = umi[1]
    editor = hostloop.UmiNameEditor(*umi) if umi else None
    ref = engines.sim_engine(p)
    want, _, _ = driver.run_engine(ref, p, fq1, fq2, pack=1000, stride=abi.qual_stride(48), want_failed=True, want_unpaired=paired,
                                   umi=editor)
    ref.close()
    g = engines.sim_engine(p)
    rc, got, lens = format_util.run_streams(g, format_util.NumpyMem(), p, fq1, fq2, 48, True, paired, umi)
    g.close()
    assert rc == 0
    for k in format_util.STREAMS:
        w = getattr(want, k, None)
        assert got[k] == (bytes(w) if w is not None else b""), f"stream {k} differs"
    assert len(got["failed"]) > 0 and sum(len(v) for v in got.values()) > 2000


def _eval_rows(mem, eng, fq: bytes, max_len):
    """text -> fastp_gpu_parse_fastq -> packed rows in `mem` (what the Evaluator entry points read)"""
    ss, qs = abi.seq_stride(max_len), abi.qual_stride(max_len)
    cap = fq.count(b"\n") // 4 + 2
    t = mem.upload(fq, (-len(fq)) % 16 + 16)
    seq, qual = mem.alloc(cap * ss), mem.alloc(cap * qs)
    lens, loff, llen = mem.alloc(cap * 2), mem.alloc(cap * 16), mem.alloc(cap * 16)
    mem.sync()
    info = eng.parse_fastq(mem.ptr(t), len(fq), True, cap, mem.ptr(seq), mem.ptr(qual), mem.ptr(lens), mem.ptr(loff), mem.ptr(llen))
    assert info.first_bad == -1
    return seq, qual, lens, info.n_records


def _overrep_reads(n, seed, L=150, n_hot=4, with_n=True):
    """reads with planted repeats of several lengths (hot substrings of every length class) and some N"""
    rng = np.random.default_rng(seed)
    d = synth.synth_pairs(n, L=L, seed=seed, paired=False)
    seq, lens = d["seq1"].copy(), d["len1"]
    hot = [rng.integers(0, 4, size=int(k)) for k in (L - 1, 120, 60, 30, 14)[:n_hot + 1]]
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n):
        r = rng.random()
        if r < 0.45:
            h = acgt[hot[int(rng.integers(0, len(hot)))]]
            if len(h) < lens[i]:
                at = int(rng.integers(0, lens[i] - len(h) + 1))
                seq[i, at:at + len(h)] = h
        if with_n and rng.random() < 0.05:
            seq[i, int(rng.integers(0, max(1, lens[i])))] = ord("N")
    return synth.to_fastq(seq, d["qual1"], lens, 1)


def _eval_case(mk_engine, mem, n, seed, L=150, trim_tail1=0):
    from fastp_amd import hostloop
    fq = _overrep_reads(n, seed, L)
    p = abi.default_params(False, L)
    g = mk_engine(p)
    seq, qual, lens, nrec = _eval_rows(mem, g, fq, L)
    b = hostloop.parse_fastq(fq, abi.qual_stride(L))
    # Evaluator::computeSeqLen
    seqlen = g.eval_seq_len(mem.ptr(lens), nrec)
    assert seqlen == evalport.evaluate_seq_len(b)
    # computeOverRepSeq: the same sequences, the same counts, the same (std::map) order
    rc, got = g.eval_overrep(mem.ptr(seq), mem.ptr(qual), mem.ptr(lens), nrec, seqlen)
    want = evalport.evaluate_overrep_counts(b, seqlen)
    assert [s for s, _ in got] == sorted(want), (len(got), len(want))
    assert dict(got) == want
    # the ten-mer histogram of evalAdapterAndReadNum
    counts = mem.alloc(4 << 20, 0x55)
    mem.sync()
    rec = g.eval_adapter_kmers(mem.ptr(seq), mem.ptr(qual), mem.ptr(lens), nrec, trim_tail1, mem.ptr(counts))
    wc, wrec = evalport.adapter_kmer_counts(b, trim_tail1)
    assert rec == wrec
    assert np.array_equal(np.frombuffer(mem.download(counts), dtype=np.uint32)[:1 << 20], wc)
    g.close()
    return got, wc


def test_sim_evaluator_prepass_equals_port():
    """fastp_gpu_eval_seq_len / eval_overrep / eval_adapter_kmers == the Python restatement of evaluator.cpp"""
    import format_util
    got, wc = _eval_case(engines.sim_engine, format_util.NumpyMem(), 1500, 11)
    assert len(got) >= 3 and {len(s) for s, _ in got} & {148, 100, 40}
    assert int(wc.sum()) > 100000
    _eval_case(engines.sim_engine, format_util.NumpyMem(), 400, 12, L=102, trim_tail1=3)   # min(150, seqlen - 2) == 100: that step runs twice
    # too small a result buffer
    fq = _overrep_reads(600, 13)
    g = engines.sim_engine(abi.default_params(False, 150))
    mem = format_util.NumpyMem()
    seq, qual, lens, nrec = _eval_rows(mem, g, fq, 150)
    rc, ns = g.eval_overrep(mem.ptr(seq), mem.ptr(qual), mem.ptr(lens), nrec, 150, max_seqs=1, check=False)
    assert rc == abi.E_OVERFLOW and ns > 1
    g.close()


def test_sim_evaluator_prepass_short_and_empty_inputs():
    """reads shorter than the step lengths / than position 20, seq_len so small that min(150, seq_len - 2) <= 0, no reads"""
    import format_util
    from fastp_amd import hostloop
    rng = np.random.default_rng(3)
    recs = []
    for i in range(300):
        L = int(rng.integers(0, 26))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L, p=[0.3, 0.3, 0.2, 0.15, 0.05]))
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * L + b"\n")
    recs += [b"@poly%d\n" % i + b"ACGTACGTACGTAC" + b"\n+\n" + b"I" * 14 + b"\n" for i in range(600)]   # a hot 10-mer family
    fq = b"".join(recs)
    g = engines.sim_engine(abi.default_params(False, 32))
    mem = format_util.NumpyMem()
    seq, qual, lens, nrec = _eval_rows(mem, g, fq, 32)
    b = hostloop.parse_fastq(fq, abi.qual_stride(32))
    for seqlen in (1, 2, 11, 12, 14, 25):
        rc, got = g.eval_overrep(mem.ptr(seq), mem.ptr(qual), mem.ptr(lens), nrec, seqlen)
        want = evalport.evaluate_overrep_counts(b, seqlen)
        assert dict(got) == want and [s for s, _ in got] == sorted(want), seqlen
    assert any(len(s) == 10 for s, _ in got)
    counts = mem.alloc(4 << 20, 0x11)
    rec = g.eval_adapter_kmers(mem.ptr(seq), mem.ptr(qual), mem.ptr(lens), nrec, 0, mem.ptr(counts))
    wc, wrec = evalport.adapter_kmer_counts(b, 0)
    assert rec == wrec and np.array_equal(np.frombuffer(mem.download(counts), dtype=np.uint32)[:1 << 20], wc)
    assert g.eval_seq_len(mem.ptr(lens), 0) == 0
    rc, got = g.eval_overrep(mem.ptr(seq), mem.ptr(qual), mem.ptr(lens), 0, 25)
    assert rc == 0 and got == []
    g.close()


def _inflate(eng, mem, comp: bytes, check_crc=True, max_blocks=100000, check=True):
    """BGZF bytes -> text through fastp_gpu_bgzf_index (host) + fastp_gpu_inflate_bgzf (device)"""
    host = np.frombuffer(comp, dtype=np.uint8)
    info, poff, plen, isz, crc, ooff = eng.bgzf_index(host, max_blocks, 1 << 40, check=check)
    d_comp = mem.upload(comp, 16)
    dev = [mem.upload(a.tobytes(), 16) for a in (poff, plen, isz, crc, ooff)]
    out = mem.alloc(max(16, int(info.out_bytes)), 0xEE)
    mem.sync()
    rc, bad = eng.inflate_bgzf(mem.ptr(d_comp), info.n_blocks, *[mem.ptr(x) for x in dev], mem.ptr(out), int(info.out_bytes),
                               check_crc, check=check)
    return info, rc, bad, mem.download(out, int(info.out_bytes))


def _se_fastq_text(n, seed):
    d = synth.synth_pairs(n, L=150, seed=seed, paired=False)
    return synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)


@pytest.mark.parametrize("level,strategy", [(6, "default"), (1, "default"), (9, "default"), (0, "default"), (6, "fixed"),
                                            (6, "huffman"), (6, "rle")])
def test_sim_bgzf_inflate_equals_zlib(level, strategy):
    """every DEFLATE block type (stored, fixed, dynamic) and match style, against the text zlib compressed"""
    import zlib
    import bgzf_util
    import format_util
    strat = {"default": zlib.Z_DEFAULT_STRATEGY, "fixed": zlib.Z_FIXED, "huffman": zlib.Z_HUFFMAN_ONLY, "rle": zlib.Z_RLE}[strategy]
    text = _se_fastq_text(700, 3)
    comp = bgzf_util.compress(text, block_bytes=20000, level=level, strategy=strat)
    g = engines.sim_engine(abi.default_params(False, 150))
    info, rc, bad, got = _inflate(g, format_util.NumpyMem(), comp)
    g.close()
    assert rc == 0 and bad == -1 and info.consumed == len(comp) and info.out_bytes == len(text)
    assert got == text


def test_sim_bgzf_inflate_lane_variant(monkeypatch):
    """FASTP_GPU_INFLATE=lane: the one-lane-per-block kernel (fq_inflate.h) stays selectable and correct"""
    import bgzf_util
    import format_util
    monkeypatch.setenv("FASTP_GPU_INFLATE", "lane")
    text = _se_fastq_text(900, 4)
    comp = bgzf_util.compress(text, block_bytes=20000, level=6)
    g = engines.sim_engine(abi.default_params(False, 150))
    info, rc, bad, got = _inflate(g, format_util.NumpyMem(), comp)
    g.close()
    assert rc == 0 and bad == -1 and got == text


@pytest.mark.parametrize("variant", ["wave", "lane"])
def test_sim_bgzf_inflate_members_with_several_deflate_blocks(variant, monkeypatch):
    """a member whose DEFLATE stream is a chain of blocks of every kind: dynamic, fixed (Z_FIXED pieces), stored (level 0
    pieces), empty stored blocks (sync flushes) - matches that reach back across block borders included"""
    import struct
    import zlib
    import format_util
    monkeypatch.setenv("FASTP_GPU_INFLATE", variant)
    text = _se_fastq_text(160, 9)[:50000]
    members = []
    for start in range(0, len(text), 25000):
        data = text[start:start + 25000]
        payload = b""
        c = None
        cuts = [0, 3000, 3001, 9000, 15000, 15002, len(data)]
        for k in range(len(cuts) - 1):
            level, strat = [(6, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (9, zlib.Z_DEFAULT_STRATEGY),
                            (1, zlib.Z_RLE), (6, zlib.Z_HUFFMAN_ONLY)][k]
            if c is None:
                c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strat)
            piece = data[cuts[k]:cuts[k + 1]]
            payload += c.compress(piece)
            payload += c.flush(zlib.Z_SYNC_FLUSH if k % 2 else zlib.Z_FULL_FLUSH)   # ends the block, adds an empty stored one
            # the next piece continues the same stream (window kept after a sync flush) with other parameters
            c2 = zlib.compressobj([0, 6, 9, 1, 6, 6][k], zlib.DEFLATED, -15, 9, [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_DEFAULT_STRATEGY,
                                                                                 zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_DEFAULT_STRATEGY][k],
                                  zdict=data[max(0, cuts[k + 1] - 32768):cuts[k + 1]]) if cuts[k + 1] > 0 and k + 1 < len(cuts) - 1 else None
            c = c2 if c2 is not None else c
        payload += c.flush(zlib.Z_FINISH)
        assert zlib.decompress(payload, -15) == data
        bsize = 18 + len(payload) + 8
        hdr = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        members.append(hdr + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))
    comp = b"".join(members)
    g = engines.sim_engine(abi.default_params(False, 150))
    info, rc, bad, got = _inflate(g, format_util.NumpyMem(), comp)
    g.close()
    assert rc == 0 and bad == -1 and got == text and info.n_blocks == 2


def test_sim_bgzf_index_chunks_and_errors():
    import bgzf_util
    import format_util
    text = _se_fastq_text(300, 4) + bytes(range(256)) * 40 + b"A" * 70000   # binary bytes, a long run (distance 1 copies)
    comp = bgzf_util.compress(text, block_bytes=0xff00)
    g = engines.sim_engine(abi.default_params(False, 150))
    mem = format_util.NumpyMem()
    # a chunk that ends inside a member: only the whole members are walked
    cut = len(comp) - 37
    info, *_ = g.bgzf_index(np.frombuffer(comp[:cut], dtype=np.uint8), 1000, 1 << 40)
    assert 0 < info.consumed < cut and comp[info.consumed:info.consumed + 2] == bytes([0x1f, 0x8b])
    # limits: blocks, text bytes
    info2, *_ = g.bgzf_index(np.frombuffer(comp, dtype=np.uint8), 1, 1 << 40)
    assert info2.n_blocks == 1 and info2.out_bytes == 0xff00
    info3, *_ = g.bgzf_index(np.frombuffer(comp, dtype=np.uint8), 1000, 0xff00 + 5)
    assert info3.n_blocks == 1
    info4, rc, bad, got = _inflate(g, mem, comp)
    assert rc == 0 and got == text
    # not BGZF: a plain gzip member
    import gzip
    info5, *_ = g.bgzf_index(np.frombuffer(gzip.compress(text[:1000]), dtype=np.uint8), 10, 1 << 40, check=False)
    assert info5.rc == abi.E_INVALID and info5.first_bad == 0
    # a corrupted payload byte is caught (CRC or stream error), a corrupted CRC field too
    bad_comp = bytearray(comp)
    bad_comp[18 + 200] ^= 0x55
    _, rc, bad, _ = _inflate(g, mem, bytes(bad_comp), check=False)
    assert rc == abi.E_INVALID and bad == 0
    first_len = int.from_bytes(comp[16:18], "little") + 1
    bad_comp = bytearray(comp)
    bad_comp[first_len - 8] ^= 1                                # CRC field of block 0
    _, rc, bad, _ = _inflate(g, mem, bytes(bad_comp), check=False)
    assert rc == abi.E_INVALID and bad == 0
    _, rc, bad, got = _inflate(g, mem, bytes(bad_comp), check_crc=False)
    assert rc == 0 and got == text
    g.close()


def test_sim_config5_shape_2x250_dedup_overrep():
    """BASELINE configs[4]'s option set: PE 2x250, --dedup at accuracy level 3 (four bloom buffers), overrepresentation"""
    L = 250
    p = abi.default_params(True, L)
    p.cut_right = 1
    p.dedup = 1
    p.dup_accuracy_level = 3
    d = synth.synth_pairs(600, L=L, seed=77, insert_mean=260.0, insert_sd=90.0, insert_min=30, insert_max=900,
                          dup_frac=0.25, polyx_frac=0.2)
    b1, b2 = cases._ArrayBatch(d["seq1"], d["len1"]), cases._ArrayBatch(d["seq2"], d["len2"])
    e1, e2 = evalport.evaluate_seq_len(b1), evalport.evaluate_seq_len(b2)
    abi.set_overrep(p, evalport.evaluate_overrep_seqs(b1, e1), evalport.evaluate_overrep_seqs(b2, e2), e1, e2, 5)
    ro, rg, co, cg = _both(p, d, True)
    for k in range(3):
        assert np.array_equal(ro[k], rg[k])
    assert np.array_equal(co, cg)


def _corruption_case(mk_engine, mem, n_reads, block_bytes, trials, min_errors):
    """random byte flips anywhere in the compressed chunk: an error or (CRC off) different text, never a write
    outside the output buffer, never accepted with the CRC check on"""
    import bgzf_util
    text = _se_fastq_text(n_reads, 5)
    comp0 = bgzf_util.compress(text, block_bytes=block_bytes)
    g = mk_engine(abi.default_params(False, 150))
    rng = np.random.default_rng(7)
    errors = 0
    for trial in range(trials):
        comp = bytearray(comp0)
        for _ in range(int(rng.integers(1, 4))):
            comp[int(rng.integers(18, len(comp) - 30))] ^= int(rng.integers(1, 256))
        comp = bytes(comp)
        info, poff, plen, isz, crc, ooff = g.bgzf_index(np.frombuffer(comp, dtype=np.uint8), 1000, 1 << 40, check=False)
        if info.n_blocks == 0:
            errors += 1
            continue
        d_comp = mem.upload(comp, 16)
        dev = [mem.upload(a.tobytes(), 16) for a in (poff, plen, isz, crc, ooff)]
        nout = int(info.out_bytes)
        out = mem.alloc(nout + 64, 0xEE)
        mem.sync()
        check_crc = bool(trial & 1)
        rc, bad = g.inflate_bgzf(mem.ptr(d_comp), info.n_blocks, *[mem.ptr(x) for x in dev], mem.ptr(out), nout, check_crc, check=False)
        got = mem.download(out)
        assert got[nout:nout + 64] == b"\xEE" * 64, f"trial {trial}: wrote past the output buffer"
        if rc == 0 and check_crc and info.consumed == len(comp0):
            assert got[:nout] == text[:nout], f"trial {trial}: CRC check passed on different text"
        errors += rc != 0
        if hasattr(mem, "keep"):
            mem.keep.clear()
    g.close()
    assert errors > min_errors


def test_sim_bgzf_inflate_survives_corruption():
    import format_util
    _corruption_case(engines.sim_engine, format_util.NumpyMem(), 120, 15000, 24, 12)   # the GPU suite runs 120 trials on full-size blocks


def _deflate(eng, mem, text: bytes, eof=False, cap=None):
    t = mem.upload(text, 16)
    cap = cap if cap is not None else len(text) + 31 * (len(text) // 65280 + 1) + 28
    out = mem.alloc(max(16, cap), 0xEE)
    mem.sync()
    rc, n = eng.deflate_bgzf(mem.ptr(t), len(text), mem.ptr(out), cap, eof, check=False)
    raw = mem.download(out)
    assert raw[min(n, cap):].count(b"\xEE") == len(raw) - min(n, cap), "bytes written past the reported length"
    return rc, raw[:min(n, cap)], n


def _deflate_texts(big):
    """FASTQ text, runs (distance-1 matches up to 258), incompressible bytes (stored blocks), block-boundary sizes"""
    rng = np.random.default_rng(9)
    fq = _se_fastq_text(700 if not big else 60000, 5)
    noise = rng.integers(0, 256, size=70000 if not big else 3 << 20, dtype=np.uint8).tobytes()
    texts = [fq, b"A" * 70001, noise, fq[:65280], fq[:65281], fq[:65279], b"ACGT" * 20000 + noise[:5000] + b"\n" * 300]
    texts += [fq[:k] for k in (1, 2, 3, 4, 5, 63, 64, 65, 257, 258, 259, 1000)]
    return texts


def _deflate_case(mk_engine, mem, big=False):
    import gzip
    import zlib
    g = mk_engine(abi.default_params(False, 150))
    for text in _deflate_texts(big):
        rc, comp, n = _deflate(g, mem, text, eof=True)
        assert rc == 0 and n == len(comp)
        assert gzip.decompress(comp) == text                       # any gzip reader: concatenated members
        assert comp.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
        # block by block, as a BGZF reader does: our own index + inflate with the CRC check
        info, rc2, bad, back = _inflate(g, mem, comp)
        assert rc2 == 0 and bad == -1 and back == text
        assert info.n_blocks == (len(text) + 65279) // 65280 + 1
        assert n <= len(text) + 31 * (len(text) // 65280 + 1) + 28
    # ratios: FASTQ text well under 40 %, a run of one byte under 1 %, noise at most 31 bytes per member over
    fq = _se_fastq_text(700 if not big else 60000, 5)
    rc, comp, n = _deflate(g, mem, fq)
    assert n < 0.4 * len(fq) and n < 1.15 * len(zlib.compress(fq, 1))
    rc, comp, n = _deflate(g, mem, b"G" * 200000)
    assert n < 2000
    # empty input: nothing, or just the end-of-file member
    rc, comp, n = _deflate(g, mem, b"")
    assert rc == 0 and n == 0
    rc, comp, n = _deflate(g, mem, b"", eof=True)
    assert rc == 0 and n == 28 and gzip.decompress(comp) == b""
    # too small an output buffer: the needed size comes back, nothing past the capacity is touched
    rc, comp, n = _deflate(g, mem, fq, cap=1000)
    assert rc == abi.E_OVERFLOW and n > 1000
    g.close()


def test_sim_deflate_bgzf_members_round_trip():
    """fastp_gpu_deflate_bgzf: every member inflates (gzip, zlib per block, our own BGZF inflate) to the text"""
    import format_util
    _deflate_case(engines.sim_engine, format_util.NumpyMem())


def _saturating_reads(n, L=150):
    """n identical reads: every base an A, every quality '~' (the largest character the packers take): each unit adds 1 and
    93 to the SAME cell of every cycle - what fills a workgroup's [count : 12 | quality sum : 20] cells of the Stats kernel"""
    stride = (L + 7) // 8 * 8
    seq = np.zeros((n, stride), dtype=np.uint8)
    qual = np.zeros((n, stride), dtype=np.uint8)
    seq[:, :L] = ord("A")
    qual[:, :L] = ord("~")
    return {"seq1": seq, "qual1": qual, "len1": np.full(n, L, dtype=np.int32)}


def test_sim_stats_cells_at_their_capacity(monkeypatch):
    """ONE Stats workgroup with the 4095 units its 12-bit cell counts hold, all of them in the same cells with the largest
    quality sum (93 * 4095 < 2^20), then 4096 units (the launch is cut into two workgroups' ranges)"""
    monkeypatch.setenv("FASTP_SIM_CUS", "1")
    monkeypatch.setenv("FASTP_GPU_STATS_BLOCKS_PER_CU", "1")
    p = abi.default_params(False, 150)
    p.adapter_seq_r1 = None
    p.adapter_enabled = 0
    p.dup_enabled = 0
    for n in (4095, 4096):
        d = _saturating_reads(n)
        ro, rg, co, cg = _both(p, d, False)
        assert ro[0].tobytes() == rg[0].tobytes()
        assert np.array_equal(co, cg), int((co != cg).sum())


def _filled_reads(n, L, ch, base="A"):
    stride = (L + 7) // 8 * 8
    seq = np.zeros((n, stride), dtype=np.uint8)
    qual = np.zeros((n, stride), dtype=np.uint8)
    seq[:, :L] = ord(base)
    qual[:, :L] = ord(ch)
    return {"seq1": seq, "qual1": qual, "len1": np.full(n, L, dtype=np.int32)}


def test_sim_stats_joint_table_at_its_capacity(monkeypatch):
    """Form 5 of the Stats kernel (fq_stats5.h): ONE workgroup with the 16383 units the slab's packed cells hold, every unit the
    same read - all of them in the SAME 16-bit halves of the joint table's cells ('K': the table's last quality row, 160 bases:
    every item whole, both halves of every dword), then in the packed cells of what the table has no row for ('L' and '~')"""
    monkeypatch.setenv("FASTP_SIM_CUS", "1")
    p = abi.default_params(False, 160)
    p.adapter_seq_r1 = None
    p.adapter_enabled = 0
    p.dup_enabled = 0
    for n, ch in ((16383, "K"), (16384, "K"), (16383, "L"), (16383, "~")):
        d = _filled_reads(n, 160, ch, base="G")
        ro, rg, co, cg = _both(p, d, False)
        assert ro[0].tobytes() == rg[0].tobytes()
        assert np.array_equal(co, cg), (n, ch, int((co != cg).sum()))


def quality_range_reads(n, L, seed, paired):
    """every quality character the packers take ('!' .. '~') with the table's edge ('K' | 'L') over-represented, runs of N, ragged
    lengths (reads that end inside an item, one-base reads, empty reads)"""
    rng = np.random.default_rng(seed)
    stride = (L + 7) // 8 * 8
    out = {}
    for tag in ("1", "2") if paired else ("1",):
        lens = rng.integers(0, L + 1, n).astype(np.int32)
        lens[rng.random(n) < 0.5] = L
        q = rng.integers(33, 127, (n, L))
        edge = rng.random((n, L)) < 0.3
        q[edge] = rng.integers(72, 79, int(edge.sum()))
        plain = rng.random(n) < 0.4                       # reads the fast path takes whole: Q2 .. Q41
        q[plain] = rng.integers(35, 75, (int(plain.sum()), L))
        c = rng.integers(0, 4, (n, L))
        isn = (rng.random((n, L)) < 0.01) & ~plain[:, None]
        seq = np.zeros((n, stride), dtype=np.uint8)
        qual = np.zeros((n, stride), dtype=np.uint8)
        valid = np.arange(L)[None, :] < lens[:, None]
        seq[:, :L] = np.where(valid, np.where(isn, ord("N"), np.frombuffer(b"ATCG", dtype=np.uint8)[c]), 0)
        qual[:, :L] = np.where(valid, q, 0)
        out["seq" + tag], out["qual" + tag], out["len" + tag] = seq, qual, lens
    return out


STATS5_RANGE_CASES = [(True, 150, {}), (False, 150, {"cut_right": 1}), (True, 100, {"trim_front1": 5, "trim_front2": 9, "cut_tail": 1}),
                      (False, 151, {"trim_front1": 17, "trim_tail1": 3}), (True, 165, {"cut_right": 1, "correction": 1}), (True, 76, {"dedup": 1}),
                      (True, 250, {"cut_right": 1, "trim_front2": 6}), (False, 400, {"cut_tail": 1}), (True, 203, {"cut_front": 1, "cut_right": 1})]   # (column blocks)


@pytest.mark.parametrize("k", range(len(STATS5_RANGE_CASES)))
def test_sim_stats_every_quality_character(k):
    """the Stats kernel's joint table holds '!' .. 'K'; every other character, N bases, ragged items and trimmed ranges take the
    base-by-base path: all of them at once, records + every counter against the oracle"""
    paired, L, extra = STATS5_RANGE_CASES[k]
    p = abi.default_params(paired, L)
    p.qualified_qual = 48
    p.unqualified_percent_limit = 90
    p.n_base_limit = 50
    p.length_required = 1
    for kk, v in extra.items():
        setattr(p, kk, v)
    d = quality_range_reads(3000, L, 100 + k, paired)
    ro, rg, co, cg = _both(p, d, paired)
    for i in range(3 if paired else 1):
        assert ro[i].tobytes() == rg[i].tobytes()
    assert np.array_equal(co, cg), int((co != cg).sum())


# --cut_front on the lane plan (round 6, DevParams::front_per_read): the forward quality cut on the window predicate the enabled
# right / tail cut builds (same window and quality), every read its own front: (paired, L, overrides)
CUT_FRONT_LANE = [
    (True, 150, dict(cut_front=1, cut_tail=1)),
    (True, 150, dict(cut_front=1, cut_right=1)),
    (False, 150, dict(cut_front=1)),
    (True, 150, dict(cut_front=1, cut_tail=1, cut_right=1, cut_front_window=6, cut_tail_window=9, cut_right_window=6, cut_front_quality=24,
                     cut_right_quality=24, trim_front1=2, trim_tail1=3, trim_front2=5, trim_tail2=1)),
    (True, 100, dict(cut_front=1, cut_front_window=8, cut_front_quality=28, umi_len1=4, umi_len2=6, umi_skip=1, trim_tail1=7, poly_x=1)),
    (False, 76, dict(cut_front=1, cut_right=1, cut_front_window=1, cut_right_window=1, cut_front_quality=30, cut_right_quality=30, dedup=1)),
    (True, 150, dict(cut_front=1, cut_tail=1, cut_front_window=5, cut_tail_window=5, cut_front_quality=18, cut_tail_quality=18,
                     adapter_seq_r1=b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", adapter_seq_r2=b"AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT", complexity_filter=1)),
]


@pytest.mark.parametrize("k", range(len(CUT_FRONT_LANE)))
def test_sim_cut_front_on_the_lane_plan(k):
    """Filter::trimAndCut's forward cut (filter.cpp:97-127) in the lane kernel, a front per read in the Stats kernel (form 5) and
    the POST Stats' cycles moved read by read (fq_front_stats_kernel): noisy reads - most of them have a bad first window -
    records + every counter against the oracle, and the plan IS the lane plan"""
    paired, L, kw = CUT_FRONT_LANE[k]
    p = abi.default_params(paired, L)
    if not paired:
        p.adapter_seq_r1 = None
    p.length_required = 8
    for key, v in kw.items():
        setattr(p, key, v)
    d = synth.noisy_reads(700, L=L, seed=300 + k, paired=paired)
    g = engines.sim_engine(p)
    assert g.plan() == "lane"
    g.close()
    ro, rg, co, cg = _both(p, d, paired)
    assert int((ro[0]["front"] > 0).sum()) > 50
    for i in range(3 if paired else 1):
        bad = np.nonzero(ro[i] != rg[i])[0]
        assert len(bad) == 0, f"case {k}: result {i} differs at {bad[:5]}: oracle {ro[i][bad[:3]]} device {rg[i][bad[:3]]}"
    assert np.array_equal(co, cg), int((co != cg).sum())


FASTA_LANE_LIST = [b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", b"AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT", b"CTGTCTCTTATACACATCT", b"TGGAATTCTCGGGTGCCAAGG",
                   b"AATGATACGGCGACCACCGAGATCTACACTCTTTCCCTACACGACGCTCTTCCGATCT", b"GATCGGAAGAGC"]


@pytest.mark.parametrize("paired,extra", [(True, {}), (False, {}), (True, {"adapter_seq_r1": b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", "trim_front1": 3, "cut_right": 1}),
                                          (False, {"poly_x": 1, "complexity_filter": 1, "cut_tail": 1})])
def test_sim_adapter_fasta_on_the_lane_plan(paired, extra):
    """AdapterTrimmer::trimByMultiSequences (adaptertrimmer.cpp:48-62) in the lane kernel (round 6: lists of sequences <= 64 bases):
    every sequence in turn on the shrinking read, the adapter events for the host's map replay - on reads that begin with an adapter
    carrying an inserted / a deleted base, exact adapters at negative positions and inside the read; records, events and every
    counter against the oracle, and a list with a longer sequence still takes the tile kernels"""
    p = abi.default_params(paired, 150)
    if not paired:
        p.adapter_seq_r1 = None
    for k, v in extra.items():
        setattr(p, k, v)
    abi.set_adapter_fasta(p, FASTA_LANE_LIST)
    g = engines.sim_engine(p)
    assert g.plan() == "lane"
    g.close()
    d = synth.adapter_indel_reads(600, L=150, seed=77, paired=paired)
    o = oraclelib.Oracle(p)
    g = engines.sim_engine(p)
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    ro, rg = o.process(*args), g.process(*args)
    co, cg = o.counters(), g.counters()
    eo, eg = o.last_adapter_events, g.last_adapter_events
    o.close()
    g.close()
    for i in range(3 if paired else 1):
        assert ro[i].tobytes() == rg[i].tobytes(), f"records {i} differ"
    assert len(eo) > 100 and eo.tobytes() == eg.tobytes(), (len(eo), len(eg))
    assert np.array_equal(co, cg), int((co != cg).sum())
    q = abi.default_params(paired, 150)
    abi.set_adapter_fasta(q, FASTA_LANE_LIST + [b"ACGT" * 20])
    g = engines.sim_engine(q)
    assert g.plan() != "lane"
    g.close()
