"""Differential fuzz over the option space: random (seeded) parameter blocks and read sets, device code (on the
SIMT emulator here, on the GPU under -m gpu) against the oracle - records, correction list and every counter.
The fixed cases of cases.py pin each option against the real reference; this walks their combinations."""
import numpy as np
import pytest

import cases
import evalport
import engines
import oraclelib
import synth
from fastp_amd import abi, engine, hostloop


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    paired = bool(rng.random() < 0.7)
    L = int(rng.choice([50, 76, 100, 150, 151, 250]))
    p = abi.default_params(paired, L)
    pick = lambda pr: bool(rng.random() < pr)
    if pick(0.3):
        p.trim_front1, p.trim_tail1 = int(rng.integers(0, 8)), int(rng.integers(0, 6))
        p.trim_front2, p.trim_tail2 = int(rng.integers(0, 8)), int(rng.integers(0, 6))
    if pick(0.2):
        p.max_len1 = int(rng.integers(L // 2, L + 5))
        p.max_len2 = int(rng.integers(L // 2, L + 5))
    for flag, w, q in (("cut_front", "cut_front_window", "cut_front_quality"), ("cut_tail", "cut_tail_window", "cut_tail_quality"),
                       ("cut_right", "cut_right_window", "cut_right_quality")):
        if pick(0.4):
            setattr(p, flag, 1)
            setattr(p, w, int(rng.choice([1, 2, 4, 5, 8, 17])))
            setattr(p, q, int(rng.integers(3, 31)))
    if pick(0.3):
        p.poly_g, p.poly_g_min_len = 1, int(rng.integers(5, 15))
    if pick(0.3):
        p.poly_x, p.poly_x_min_len = 1, int(rng.integers(5, 15))
    p.adapter_enabled = int(pick(0.8))
    if not paired:
        p.adapter_seq_r1 = cases.ADAPTER_R1.encode() if (p.adapter_enabled and pick(0.7)) else None
    elif p.adapter_enabled and pick(0.3):
        p.adapter_seq_r1, p.adapter_seq_r2 = cases.ADAPTER_R1.encode(), cases.ADAPTER_R2.encode()
    if paired:
        p.allow_gap_overlap_trimming = int(pick(0.25))
        p.correction = int(pick(0.4))
        if pick(0.25):
            p.merge, p.correction = 1, 1
            p.merge_include_unmerged = int(pick(0.5))
        p.overlap_require = int(rng.choice([30, 30, 20, 12]))
        p.overlap_diff_limit = int(rng.choice([5, 5, 2, 9]))
        p.overlap_diff_percent_limit = int(rng.choice([20, 20, 10, 35]))
    if pick(0.3):
        p.qualified_qual = int(rng.integers(5, 30))
        p.unqualified_percent_limit = int(rng.integers(5, 70))
        p.n_base_limit = int(rng.integers(0, 8))
        p.avg_qual_req = int(rng.choice([0, 0, 15, 25]))
    p.qual_filter = int(pick(0.85))
    p.length_filter = int(pick(0.85))
    if pick(0.3):
        p.length_required = int(rng.integers(1, L // 2))
        p.length_limit = int(rng.choice([0, L - 5, L - 20]))
    if pick(0.25):
        p.complexity_filter, p.complexity_threshold = 1, float(rng.choice([0.2, 0.3, 0.45]))
    p.dup_enabled = int(pick(0.85))
    if p.dup_enabled:
        p.dedup = int(pick(0.3))
        p.dup_accuracy_level = int(rng.choice([1, 1, 2, 3]))
    if pick(0.2) and not p.merge:
        p.umi_len1 = int(rng.integers(1, 10))
        p.umi_len2 = int(rng.integers(0, 10)) if paired else 0
        p.umi_skip = int(rng.integers(0, 4))
    n = int(rng.integers(150, 420))
    mean = float(rng.choice([0.6, 0.9, 1.3, 2.0])) * L
    d = synth.synth_pairs(n, L=L, seed=seed, paired=paired, insert_mean=mean, insert_sd=0.35 * mean, insert_min=5,
                          insert_max=max(800, 3 * L), polyg_frac=float(rng.choice([0.0, 0.2])), polyx_frac=float(rng.choice([0.0, 0.25])),
                          dup_frac=float(rng.choice([0.05, 0.3])), ragged_frac=float(rng.choice([0.0, 0.02, 0.3])),
                          lowq_site_rate=float(rng.choice([0.01, 0.03, 0.1])))
    if pick(0.3) and not (p.umi_len1 or p.umi_len2):   # overrepresentation analysis with the seeds the pre-pass would find
        b1 = cases._ArrayBatch(d["seq1"], d["len1"])
        e1 = evalport.evaluate_seq_len(b1)
        s1 = evalport.evaluate_overrep_seqs(b1, e1)
        e2, s2 = 0, []
        if paired:
            b2 = cases._ArrayBatch(d["seq2"], d["len2"])
            e2 = evalport.evaluate_seq_len(b2)
            s2 = evalport.evaluate_overrep_seqs(b2, e2)
        abi.set_overrep(p, s1, s2, e1, e2, int(rng.choice([1, 2, 7, 20])))
    if pick(0.15) and p.adapter_enabled:
        abi.set_adapter_fasta(p, [b"CTGTCTCTTATACACATCT", b"AGATCGGAAGAGC", b"TGGAATTCTCGGGTGCCAAGG"][:int(rng.integers(1, 4))])
    # (drawn last so that the earlier seeds keep their cases) --overlapped_out; adapters longer than 64 bases
    if paired and pick(0.2):   # with merge as well (the records' reserved fields go to --overlapped_out then)
        p.overlapped_out = 1
    if p.adapter_enabled and p.adapter_seq_r1 and pick(0.15):
        p.adapter_seq_r1 = cases.LONG_R1.encode()
        if paired:
            p.adapter_seq_r2 = cases.LONG_R2.encode()
    # letters outside ACGTN (soft-masked stretches, IUPAC codes, '.'): the text kernel (fq_text.h) takes those units
    # (with -p the seeds were evaluated on the reads as they were before: seeds are ACGTN, fq_host.cpp refuses others)
    if pick(0.3):
        synth.add_exotic(d, seed=seed, read_frac=float(rng.choice([0.005, 0.05, 0.4])), paired=paired)
    return p, d, paired


def _check(mk_engine, seed):
    p, d, paired = random_case(seed)
    args = (d["seq1"], d["qual1"], d["len1"]) + ((d["seq2"], d["qual2"], d["len2"]) if paired else ())
    o = oraclelib.Oracle(p)
    try:
        g = mk_engine(p)
    except engine.EngineError as e:
        o.close()
        assert e.code == abi.E_UNSUPPORTED, f"seed {seed}: {e}"
        pytest.skip(f"seed {seed}: combination refused by the device path ({e})")
    ro, rg = o.process(*args), g.process(*args)
    co, cg = o.counters(), g.counters()
    evo, evg = getattr(o, "last_adapter_events", None), getattr(g, "last_adapter_events", None)
    o.close()
    g.close()
    for k, what in enumerate(("r1", "r2", "pair")):
        if ro[k] is not None:
            bad = np.nonzero(ro[k] != rg[k])[0]
            assert len(bad) == 0, f"seed {seed}: {what} differs at {bad[:5]}: oracle {ro[k][bad[:3]]} device {rg[k][bad[:3]]}"
    assert np.array_equal(np.sort(ro[3], order=["read", "pos"]), np.sort(rg[3], order=["read", "pos"])), f"seed {seed}: corrections differ"
    bad = np.nonzero(co != cg)[0]
    assert len(bad) == 0, f"seed {seed}: {len(bad)} counters differ, first at {bad[:6]}"
    if evo is not None and evg is not None:
        assert np.array_equal(evo, evg), f"seed {seed}: adapter events differ"


@pytest.mark.parametrize("seed", range(40))
def test_sim_random_option_sets_equal_oracle(seed):
    _check(engines.sim_engine, seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(40, 85))
def test_gpu_random_option_sets_equal_oracle(seed):
    _check(engines.gpu_engine, seed)
