"""Count the inputs on which the reference's one-gap code paths ACCEPT (test infrastructure): the parity tests for
Matcher::matchWithOneInsertion / diffWithOneInsertion assert these counts before comparing, so that a passing
comparison cannot mean "both sides said no"."""
import numpy as np

import oraclelib
from fastp_amd import abi


def gap_overlap_pairs(d, params, limit=None):
    """pairs for which OverlapAnalysis::analyze(allowGap) returns hasGap (overlapanalysis.cpp:91-139), via the
    oracle's string-level entry point: (count, count with offset < 0)"""
    L = oraclelib.lib()
    n = len(d["len1"]) if limit is None else min(limit, len(d["len1"]))
    tot = neg = 0
    for i in range(n):
        l1, l2 = int(d["len1"][i]), int(d["len2"][i])
        ov = L.fastp_oracle_analyze(d["seq1"][i, :l1].tobytes(), l1, d["seq2"][i, :l2].tobytes(), l2,
                                    int(params.overlap_diff_limit), int(params.overlap_require),
                                    float(params.overlap_diff_percent_limit) / 100.0, 1)
        if ov.has_gap:
            tot += 1
            neg += ov.offset < 0
    return tot, neg


def gap_trimmed_pairs(r1, pr):
    """pairs trimmed by trimByOverlapAnalysis although the (ungapped) OverlapResult of the record says "not
    overlapped": the one-gap result was used (peprocessor.cpp:445-447)"""
    return int((((r1["flags"] & abi.RF_ADAPTER_OV) != 0) & ((pr["flags"] & abi.PF_OVERLAPPED) == 0)).sum())


def gap_adapter_trims(seq, lens, res, adapter):
    """reads cut by AdapterTrimmer::trimBySequence whose cut position fails the Hamming test of the first loop
    (adaptertrimmer.cpp:87-100): the match came from the one-insertion / one-deletion loops (:105-135)"""
    a = np.frombuffer(adapter, dtype=np.uint8)
    alen = len(a)
    cnt = 0
    idx = np.nonzero(((res["flags"] & abi.RF_ADAPTER) != 0) & ((res["flags"] & abi.RF_ADAPTER_OV) == 0))[0]
    for i in idx:
        pos = int(res["adapter_pos"][i])
        rlen = int(res["len"][i]) + int(res["adapter_len"][i]) if pos >= 0 else None
        if pos < 0:
            continue   # negative positions only come from the Hamming loop
        r = seq[i, :rlen]
        cmplen = min(rlen - pos, alen)
        mm = int((r[pos:pos + cmplen] != a[:cmplen]).sum())
        if mm > cmplen // 8:
            cnt += 1
    return cnt
