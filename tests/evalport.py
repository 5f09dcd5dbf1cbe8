"""evalport.py - TEST INFRASTRUCTURE: a Python restatement of the counting loops of the reference's Evaluator
pre-pass (/root/reference/src/evaluator.cpp), the checker for fastp_gpu_eval_* (include/fastp_gpu.h).
evaluate_overrep_seqs is pinned by the pe_overrep / se_overrep goldens: the seed set it returns is what the
golden runs' JSON (written by fastp_ref) was reproduced from."""
import numpy as np


def evaluate_seq_len(b) -> int:  # Evaluator::computeSeqLen evaluator.cpp:54-76
    n = min(b.n, 1000)
    return int(b.lens[:n].max()) if n else 0


def evaluate_overrep_seqs(b, seqlen: int) -> list[bytes]:
    """Evaluator::computeOverRepSeq (evaluator.cpp:78-169): hot substrings of the first ~1.5 Mbases,
    minus those that are substrings of a hotter/longer one; returned in std::map (sorted) order"""
    BASE_LIMIT = 151 * 10000
    counts: dict[bytes, int] = {}
    bases = 0
    i = 0
    steps = [10, 20, 40, 100, min(150, seqlen - 2)]
    while bases < BASE_LIMIT and i < b.n:
        rlen = int(b.lens[i])
        seq = b.seq[i, :rlen].tobytes()
        bases += rlen
        for step in steps:
            if step <= 0:
                continue
            for k in range(0, rlen - step):
                sub = seq[k:k + step]
                counts[sub] = counts.get(sub, 0) + 1
        i += 1
    hot: dict[bytes, int] = {}
    for seq, c in counts.items():
        L = len(seq)
        if L >= seqlen - 1:
            ok = c >= 3
        elif L >= 100:
            ok = c >= 5
        elif L >= 40:
            ok = c >= 20
        elif L >= 20:
            ok = c >= 100
        elif L >= 10:
            ok = c >= 500
        else:
            ok = False
        if ok:
            hot[seq] = c
    keys = sorted(hot)
    removed = set()
    for seq in keys:  # :140-160: erase while iterating; later comparisons see the shrunken map
        c = hot[seq]
        for seq2 in keys:
            if seq2 in removed or seq2 == seq:
                continue
            if seq in seq2 and c // hot[seq2] < 10:
                removed.add(seq)
                break
    return [k for k in keys if k not in removed]


def evaluate_overrep_counts(b, seqlen: int) -> dict:
    """the same, with the counts kept ({sequence: count}, removal applied)"""
    keep = evaluate_overrep_seqs(b, seqlen)
    BASE_LIMIT = 151 * 10000
    want = set(keep)
    counts = {k: 0 for k in keep}
    bases = 0
    i = 0
    steps = [10, 20, 40, 100, min(150, seqlen - 2)]
    while bases < BASE_LIMIT and i < b.n:
        rlen = int(b.lens[i])
        seq = b.seq[i, :rlen].tobytes()
        bases += rlen
        for step in steps:
            if step <= 0:
                continue
            for k in range(0, rlen - step):
                sub = seq[k:k + step]
                if sub in want:
                    counts[sub] += 1
        i += 1
    return counts


def adapter_kmer_counts(b, trim_tail1: int = 0) -> tuple[np.ndarray, int]:
    """the 4^10 ten-mer histogram of Evaluator::evalAdapterAndReadNum (evaluator.cpp:313-341 loading limits,
    :377-402 counting): returns (counts[1 << 20] uint32, records loaded)"""
    READ_LIMIT = 256 * 1024
    BASE_LIMIT = 151 * READ_LIMIT
    code = np.full(256, -1, dtype=np.int64)
    for ch, v in ((b"A", 0), (b"T", 1), (b"C", 2), (b"G", 3)):  # Evaluator::seq2int :573-625
        code[ch[0]] = v
    counts = np.zeros(1 << 20, dtype=np.int64)
    shift_tail = max(1, trim_tail1)
    records = 0
    bases = 0
    while records < READ_LIMIT and bases < BASE_LIMIT and records < b.n:
        rlen = int(b.lens[records])
        bases += rlen
        c = code[b.seq[records, :rlen]]
        records += 1
        last = rlen - 10 - shift_tail
        if last < 20:
            continue
        npos = last - 20 + 1
        key = np.zeros(npos, dtype=np.int64)
        bad = np.zeros(npos, dtype=bool)
        for k in range(10):
            w = c[20 + k:20 + k + npos]
            bad |= w < 0
            key = (key << 2) | np.where(w < 0, 0, w)
        np.add.at(counts, key[~bad], 1)
    counts[0] = 0
    return counts.astype(np.uint32), records
