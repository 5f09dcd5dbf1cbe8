"""Known-answer tests the reference carries for the hot path, replayed on the oracle.

Vectors: Filter::test (filter.cpp:245-264), OverlapAnalysis::test
(overlapanalysis.cpp:181-210), AdapterTrimmer::test (adaptertrimmer.cpp:159-184),
BaseCorrector::test (basecorrector.cpp:85-107), PolyX::test (polyx.cpp:118-130),
fastp_simd::testSimd semantics (simd.cpp:326-564, via the functions that use them).
"""
import ctypes as C

import numpy as np

import oraclelib
from fastp_amd import abi

L = oraclelib.lib


def test_filter_trim_and_cut_kat():
    p = abi.default_params(False, 64)
    p.cut_front = 1
    p.cut_tail = 1
    p.cut_front_window = p.cut_tail_window = 4
    p.cut_front_quality = p.cut_tail_quality = 20
    seq = b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTT"
    qual = b"/////CCCCCCCCCCCC////CCCCCCCCCCCCCC////E"
    f, n = C.c_int(), C.c_int()
    ok = L().fastp_oracle_trim_and_cut(C.byref(p), seq, qual, len(seq), 0, 1, C.byref(f), C.byref(n))
    assert ok == 1
    assert seq[f.value:f.value + n.value] == b"CCCCCCCCCCCCCCCCCCCCCCCCCCCC"
    assert qual[f.value:f.value + n.value] == b"CCCCCCCCCCC////CCCCCCCCCCCCC"


def test_overlap_analysis_kat():
    r1 = b"CAGCGCCTACGGGCCCCTTTTTCTGCGCGACCGCGTGGCTGTGGGCGCGGATGCCTTTGAGCGCGGTGACTTCTCACTGCGTATCGAGC"
    r2 = b"ACCTCCAGCGGCTCGATACGCAGTGAGAAGTCACCGCGCTCAAAGGCATCCGCGCCCACAGCCACGCGGTCGCGCAGAAAAAGGGGTCC"
    ov = L().fastp_oracle_analyze(r1, len(r1), r2, len(r2), 2, 30, 0.2, 0)
    assert (ov.overlapped, ov.offset, ov.overlap_len, ov.diff) == (1, 10, 79, 1)
    # late-mismatch case: only the first 50 bases are bounded, diff is the full count
    a = b"A" * 50 + b"C" * 30
    rc = b"A" * 50 + b"G" * 30
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    r2b = rc.translate(comp)[::-1]
    ov = L().fastp_oracle_analyze(a, len(a), r2b, len(r2b), 0, 30, 0.0, 0)
    assert (ov.overlapped, ov.offset, ov.overlap_len, ov.diff) == (1, 0, 80, 30)


def test_adapter_trimmer_kat():
    seq = b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAATTTTCCCCGGGG"
    adapter = b"TTTTCCACGGGGATACTACTG"
    pos = C.c_int()
    found = L().fastp_oracle_trim_by_sequence(seq, len(seq), adapter, len(adapter), 4, C.byref(pos))
    assert found == 1
    assert seq[:pos.value] == b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAA"
    # trimByMultiSequences (adaptertrimmer.cpp:48-62): matchReq 4 for <=16 adapters
    read = (b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAATTTTCCCCGGGGAAATTTCCCGGGAAATTTCCCGGG"
            b"ATCGATCGATCGATCGAATTCC")
    cur = read
    for a in (b"GCTAGCTAGCTAGCTA", b"AAATTTCCCGGGAAATTTCCCGGG", b"ATCGATCGATCGATCG", b"AATTCCGGAATTCCGG"):
        if L().fastp_oracle_trim_by_sequence(cur, len(cur), a, len(a), 4, C.byref(pos)):
            cur = cur[:max(pos.value, 0)]
    assert cur == b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAATTTTCCCCGGGG"


def test_polyx_kat():
    seq = b"ATTTTAAAAAAAAAATAAAAAAAAAAAAACAAAAAAAAAAAAAAAAAAAAAAAAAT"
    base, trimmed = C.c_int(), C.c_int()
    n = L().fastp_oracle_trim_poly_x(seq, len(seq), 10, C.byref(base), C.byref(trimmed))
    assert seq[:n] == b"ATTTT" and trimmed.value == 51 and base.value == 0


def _stride_rows(seqs, stride):
    n = len(seqs)
    a = np.zeros((n, stride), dtype=np.uint8)
    for i, s in enumerate(seqs):
        a[i, :len(s)] = np.frombuffer(s, dtype=np.uint8)
    return a, np.array([len(s) for s in seqs], dtype=np.int32)


def test_base_corrector_kat():
    p = abi.default_params(True, 64)
    p.correction = 1
    p.adapter_enabled = 0
    p.qual_filter = 0
    p.length_filter = 0
    p.dup_enabled = 0
    s1 = b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAATTTTCCACGGGG"
    q1 = b"EEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEE/EEEEE"
    s2 = b"AAAAAAAAAACCCCGGGGAAAATTTTAAAATTGGGGGGGGGGTGGGGGGGGGGGGG"
    q2 = b"EEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEEE/EEEEEEEEEEEEE"
    eng = oraclelib.Oracle(p)
    S1, L1 = _stride_rows([s1], 64)
    Q1, _ = _stride_rows([q1], 64)
    S2, L2 = _stride_rows([s2], 64)
    Q2, _ = _stride_rows([q2], 64)
    r1, r2, pr, corr = eng.process(S1, Q1, L1, S2, Q2, L2)
    eng.close()
    o1, oq1, o2, oq2 = bytearray(s1), bytearray(q1), bytearray(s2), bytearray(q2)
    for c in corr:
        tgt, tq = (o1, oq1) if c["read"] % 2 == 0 else (o2, oq2)
        tgt[c["pos"]] = c["base"]
        tq[c["pos"]] = c["qual"]
    assert bytes(o1) == b"TTTTAACCCCCCCCCCCCCCCCCCCCCCCCCCCCAATTTTAAAATTTTCCCCGGGG"
    assert bytes(o2) == b"AAAAAAAAAACCCCGGGGAAAATTTTAAAATTGGGGGGGGGGGGGGGGGGGGGGGG"
    assert bytes(oq1) == b"E" * len(q1) and bytes(oq2) == b"E" * len(q2)


def test_analyze_empty_reads_quirk():
    """a zero-length mate is 'overlapped' with overlap_len 0 (overlapanalysis.cpp:48-64 with
    len2 == 0 accepts at offset 0) - exercised by the reference's testdata (empty R1 record)"""
    ov = L().fastp_oracle_analyze(b"A" * 60, 60, b"", 0, 5, 30, 0.2, 0)
    assert (ov.overlapped, ov.offset, ov.overlap_len, ov.diff) == (1, 0, 0, 0)
    ov = L().fastp_oracle_analyze(b"", 0, b"A" * 60, 60, 5, 30, 0.2, 0)
    assert (ov.overlapped, ov.offset, ov.overlap_len, ov.diff) == (1, 0, 0, 0)
    ov = L().fastp_oracle_analyze(b"A" * 20, 20, b"T" * 20, 20, 5, 30, 0.2, 0)
    assert ov.overlapped == 0


def test_dup_hash_is_positional():
    out = (C.c_uint64 * 8)()
    n = L().fastp_oracle_dup_hash(1, b"ACGT", 4, None, 0, out)
    assert n == 2
    # duplicate.cpp:111-120 by hand: primes start 10007, 20011(first prime >=20008) ...
    vals = {"A": 7, "T": 222, "C": 74, "G": 31}

    def primes(k):
        res, num = [], 10000
        while len(res) < k:
            num += 1
            if all(num % d for d in range(2, int(num ** 0.5) + 1)):
                res.append(num)
                num += 10000
        return res
    P = primes(8)
    h0 = sum(P[(p * 2 + 0) & 1023] * (vals[c] + p) for p, c in enumerate("ACGT")) % (1 << 64)
    h1 = sum(P[(p * 2 + 1) & 1023] * (vals[c] + p) for p, c in enumerate("ACGT")) % (1 << 64)
    assert (out[0], out[1]) == (h0, h1)
