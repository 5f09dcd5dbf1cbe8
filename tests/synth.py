"""Deterministic synthetic FASTQ (fragment model, SURVEY.md 8d "S-B") - test/bench input.

Not a restatement of anything in the reference: it only has to produce reads that
exercise every branch of the per-read path (adapter read-through, overlapping and
non-overlapping pairs, decaying quality, N bases, polyG/polyA tails, duplicates,
ragged lengths incl. 0/1/29/30/31).
"""
import numpy as np

ADAPTER_R1 = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"   # Illumina TruSeq Adapter Read 1
ADAPTER_R2 = b"AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"   # Illumina TruSeq Adapter Read 2
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP_CODE = np.array([3, 2, 1, 0], dtype=np.uint8)  # A<->T, C<->G on ACGT codes


def _adapter_pad(adapter, L):
    a = np.frombuffer(adapter, dtype=np.uint8)
    return np.concatenate([a, np.full(L + 8, ord("A"), dtype=np.uint8)])


def synth_pairs(n, L=150, seed=42, insert_mean=300.0, insert_sd=80.0, insert_min=20, insert_max=800,
                polyg_frac=0.0, polyx_frac=0.0, dup_frac=0.10, ragged_frac=0.02, n_rate=0.5,
                lowq_site_rate=0.03, paired=True, gen=None, adapters=None, exotic_frac=0.0):
    """returns dict(seq1,qual1,len1[,seq2,qual2,len2]) as ASCII uint8 [n, stride] + int32 lens"""
    if exotic_frac > 0:   # letters outside ACGTN on top of whatever the other arguments produce
        d = synth_pairs(n, L=L, seed=seed, insert_mean=insert_mean, insert_sd=insert_sd, insert_min=insert_min, insert_max=insert_max,
                        polyg_frac=polyg_frac, polyx_frac=polyx_frac, dup_frac=dup_frac, ragged_frac=ragged_frac, n_rate=n_rate,
                        lowq_site_rate=lowq_site_rate, paired=paired, gen=gen, adapters=adapters)
        return add_exotic(d, seed=seed, read_frac=exotic_frac, paired=paired)
    if gen == "indel_overlap":   # parity cases that need single-base indels (the one-gap ACCEPT paths)
        return indel_overlap_pairs(n, L=L, seed=seed)
    if gen == "adapter_indel":
        return adapter_indel_reads(n, L=L, seed=seed, paired=paired, adapters=adapters)
    if gen == "late_long":   # nothing longer than 100 bases among the first 1100 units: what Evaluator::computeSeqLen sees (the first 1000 reads)
        d = synth_pairs(n, L=L, seed=seed, insert_mean=insert_mean, insert_sd=insert_sd, paired=paired)
        head = min(1100, (2 * n) // 3)   # (inputs of fewer than 1650 units: two thirds of them)
        for m in ("1", "2") if paired else ("1",):
            d["len" + m][:head] = np.minimum(d["len" + m][:head], 100)
        assert n < 2 or int(d["len1"][head:].max()) > 100
        return d
    rng = np.random.default_rng(seed)
    stride = (L + 7) // 8 * 8
    ins = np.clip(np.rint(rng.normal(insert_mean, insert_sd, n)), insert_min, insert_max).astype(np.int64)
    F = rng.integers(0, 4, size=(n, insert_max), dtype=np.uint8)   # fragment codes
    # exact duplicates: copy an earlier fragment + insert size
    ndup = int(n * dup_frac)
    if ndup and n > 1:
        dst = rng.choice(np.arange(1, n), size=min(ndup, n - 1), replace=False)
        src = (rng.random(len(dst)) * dst).astype(np.int64)
        F[dst] = F[src]
        ins[dst] = ins[src]
    j = np.arange(L)[None, :]

    def mate(codes_at, adapter):
        pad = _adapter_pad(adapter, L)
        inside = j < ins[:, None]
        seq = np.where(inside, _ACGT[codes_at], pad[np.clip(j - ins[:, None], 0, len(pad) - 1)])
        return seq.astype(np.uint8)

    idx1 = np.clip(j, 0, insert_max - 1) + np.zeros((n, 1), dtype=np.int64)
    s1 = mate(np.take_along_axis(F, idx1, axis=1), ADAPTER_R1)
    outs = {}
    mates = [("1", s1)]
    if paired:
        idx2 = np.clip(ins[:, None] - 1 - j, 0, insert_max - 1)
        s2 = mate(_COMP_CODE[np.take_along_axis(F, idx2, axis=1)], ADAPTER_R2)
        mates.append(("2", s2))
    for tag, s in mates:
        # quality: Q37 plateau, drop point in [L/2, L+40) to Q2..24, 3% random low-Q sites
        drop = rng.integers(L // 2, L + 40, size=n)
        lowq = rng.integers(2, 25, size=n)
        q = np.where(j < drop[:, None], 37, lowq[:, None]).astype(np.int64)
        site = rng.random((n, L)) < lowq_site_rate
        q = np.where(site, rng.integers(2, 30, size=(n, L)), q)
        # substitution errors at 10^(-Q/10)
        perr = np.power(10.0, -q / 10.0)
        e = rng.random((n, L)) < perr
        sub = _ACGT[rng.integers(0, 4, size=(n, L))]
        s = np.where(e, sub, s)
        # Q<=2 sites become N with probability n_rate
        isn = (q <= 2) & (rng.random((n, L)) < n_rate)
        s = np.where(isn, ord("N"), s).astype(np.uint8)
        # polyG / polyX tails
        if polyg_frac > 0:
            pg = rng.random(n) < polyg_frac
            tl = rng.integers(8, 60, size=n)
            tail = (j >= (L - tl)[:, None]) & pg[:, None]
            keep = rng.random((n, L)) < 0.03   # a few non-G inside the tail
            s = np.where(tail & ~keep, ord("G"), s).astype(np.uint8)
        if polyx_frac > 0:
            px = rng.random(n) < polyx_frac
            tl = rng.integers(8, 60, size=n)
            base = _ACGT[rng.integers(0, 4, size=n)]
            tail = (j >= (L - tl)[:, None]) & px[:, None]
            keep = rng.random((n, L)) < 0.03
            s = np.where(tail & ~keep, base[:, None], s).astype(np.uint8)
        lens = np.full(n, L, dtype=np.int32)
        if ragged_frac > 0:
            rg = rng.random(n) < ragged_frac
            special = np.array([0, 1, 2, 14, 15, 16, 29, 30, 31, 32, 49, 50, 51, L - 1], dtype=np.int32)
            pick = np.where(rng.random(n) < 0.5, special[rng.integers(0, len(special), size=n)],
                            rng.integers(0, L + 1, size=n).astype(np.int32))
            lens = np.where(rg, np.minimum(pick, L), lens).astype(np.int32)
        seq = np.zeros((n, stride), dtype=np.uint8)
        qual = np.zeros((n, stride), dtype=np.uint8)
        valid = j < lens[:, None]
        seq[:, :L] = np.where(valid, s, 0)
        qual[:, :L] = np.where(valid, q + 33, 0)
        outs["seq" + tag] = seq
        outs["qual" + tag] = qual
        outs["len" + tag] = lens
    return outs


def add_exotic(d, seed=1, read_frac=0.12, paired=True):
    """Letters outside ACGTN in a fraction of the reads (in place): soft-masked (lower-case) stretches and whole reads,
    IUPAC codes, '.' - what the reference bins by `base & 7` (stats.cpp:206-208), hashes as 13, complements to N ..."""
    rng = np.random.default_rng(seed + 977)
    iupac = np.frombuffer(b"RYKMSWBDHVryn.uX-", dtype=np.uint8)
    for tag in ("1", "2") if paired else ("1",):
        seq, lens = d["seq" + tag], d["len" + tag]
        for i in np.flatnonzero(rng.random(len(lens)) < read_frac):
            L = int(lens[i])
            if L == 0:
                continue
            kind = int(rng.integers(0, 5))
            if kind == 0:     # a soft-masked stretch
                a = int(rng.integers(0, L))
                b = min(L, a + int(rng.integers(1, 60)))
                seq[i, a:b] |= 0x20
            elif kind == 1:   # the whole read in lower case
                seq[i, :L] |= 0x20
            elif kind == 2:   # a few IUPAC / foreign letters
                pos = rng.integers(0, L, size=int(rng.integers(1, 6)))
                seq[i, pos] = iupac[rng.integers(0, len(iupac), size=len(pos))]
            elif kind == 3:   # '.' for no-calls, also where an N was
                pos = np.flatnonzero(seq[i, :L] == ord("N"))
                if len(pos) == 0:
                    pos = rng.integers(0, L, size=2)
                seq[i, pos] = ord(".")
            else:             # a foreign letter at the ends (trimming, polyX, adapter matching look there first)
                seq[i, L - 1] = iupac[int(rng.integers(0, len(iupac)))]
                if rng.random() < 0.5:
                    seq[i, 0] = iupac[int(rng.integers(0, len(iupac)))]
    return d


def noisy_reads(n, L=150, seed=1, paired=True, n_rate=0.06, qlo=2, qhi=41):
    """adversarial input for the quality-cutting / N-skipping scans: i.i.d. qualities with runs of
    good and bad bases, frequent N (also at both ends), every length 0..L"""
    rng = np.random.default_rng(seed)
    stride = (L + 7) // 8 * 8
    j = np.arange(L)[None, :]
    outs = {}
    for tag in ("1", "2") if paired else ("1",):
        lens = rng.integers(0, L + 1, size=n).astype(np.int32)
        lens[rng.random(n) < 0.5] = L
        # piecewise quality: runs of 1..12 bases share a level, plus single-base noise
        runs = rng.integers(qlo, qhi, size=(n, L))
        keep = rng.random((n, L)) < 0.25
        keep[:, 0] = True
        idx = np.maximum.accumulate(np.where(keep, j, 0), axis=1)
        q = np.take_along_axis(runs, idx, axis=1)
        noise = rng.random((n, L)) < 0.1
        q = np.where(noise, rng.integers(qlo, qhi, size=(n, L)), q)
        s = _ACGT[rng.integers(0, 4, size=(n, L))]
        isn = rng.random((n, L)) < n_rate
        edge = (rng.random((n, 1)) < 0.3) & ((j < rng.integers(0, 6, size=(n, 1))) | (j >= lens[:, None] - rng.integers(0, 6, size=(n, 1))))
        s = np.where(isn | edge, ord("N"), s).astype(np.uint8)
        valid = j < lens[:, None]
        seq = np.zeros((n, stride), dtype=np.uint8)
        qual = np.zeros((n, stride), dtype=np.uint8)
        seq[:, :L] = np.where(valid, s, 0)
        qual[:, :L] = np.where(valid, q + 33, 0)
        outs["seq" + tag], outs["qual" + tag], outs["len" + tag] = seq, qual, lens
    return outs


def overlap_pairs(n, L=150, seed=1, err=0.03, n_rate=0.02):
    """adversarial input for OverlapAnalysis::analyze: every insert size from 1 to 2L (read-through
    on both sides and barely-overlapping pairs), both mates cut to independent random lengths,
    substitutions at `err`, N on both strands, a burst of mismatches after base 50 in some pairs"""
    rng = np.random.default_rng(seed)
    stride = (L + 7) // 8 * 8
    j = np.arange(L)[None, :]
    ins = rng.integers(1, 2 * L + 1, size=n)
    F = rng.integers(0, 4, size=(n, 2 * L + 1), dtype=np.uint8)
    rnd1 = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    rnd2 = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    c1 = np.where(j < ins[:, None], np.take_along_axis(F, np.minimum(j, 2 * L) + np.zeros((n, 1), dtype=np.int64), axis=1), rnd1)
    i2 = np.clip(ins[:, None] - 1 - j, 0, 2 * L)
    c2 = np.where(j < ins[:, None], _COMP_CODE[np.take_along_axis(F, i2, axis=1)], rnd2)
    outs = {}
    for tag, c in (("1", c1), ("2", c2)):
        e = rng.random((n, L)) < err * rng.random((n, 1)) * 2
        burst = (rng.random((n, 1)) < 0.15) & (j >= 50) & (j < 50 + rng.integers(1, 40, size=(n, 1)))
        c = np.where(e | (burst & (rng.random((n, L)) < 0.5)), rng.integers(0, 4, size=(n, L)), c)
        s = _ACGT[c]
        s = np.where(rng.random((n, L)) < n_rate * (rng.random((n, 1)) < 0.5), ord("N"), s).astype(np.uint8)
        lens = np.where(rng.random(n) < 0.6, L, rng.integers(0, L + 1, size=n)).astype(np.int32)
        q = np.where(rng.random((n, L)) < 0.1, rng.integers(2, 15, size=(n, L)), rng.integers(30, 41, size=(n, L)))
        valid = j < lens[:, None]
        seq = np.zeros((n, stride), dtype=np.uint8)
        qual = np.zeros((n, stride), dtype=np.uint8)
        seq[:, :L] = np.where(valid, s, 0)
        qual[:, :L] = np.where(valid, q + 33, 0)
        outs["seq" + tag], outs["qual" + tag], outs["len" + tag] = seq, qual, lens
    return outs


def _finish(c, lens, q, L):
    """ACGT codes [n, L] + lengths + phred qualities -> padded ASCII rows"""
    n = len(lens)
    stride = (L + 7) // 8 * 8
    j = np.arange(L)[None, :]
    valid = j < lens[:, None]
    seq = np.zeros((n, stride), dtype=np.uint8)
    qual = np.zeros((n, stride), dtype=np.uint8)
    seq[:, :L] = np.where(valid, _ACGT[c], 0)
    qual[:, :L] = np.where(valid, q + 33, 0)
    return seq, qual, lens.astype(np.int32)


def _indel_rows(c, kind, pos, rng):
    """one-base insertion (kind 1: a random base enters at pos, the tail moves right and loses its last base) or
    deletion (kind 2: base pos leaves, the tail moves left, a random base fills the end) per row; kind 0: untouched"""
    n, L = c.shape
    j = np.arange(L)[None, :]
    p = pos[:, None]
    src_ins = np.where(j > p, j - 1, j)
    src_del = np.where(j >= p, np.minimum(j + 1, L - 1), j)
    ins = np.take_along_axis(c, src_ins, axis=1)
    ins = np.where(j == p, rng.integers(0, 4, size=(n, 1)), ins)
    dele = np.take_along_axis(c, src_del, axis=1)
    dele[:, L - 1] = rng.integers(0, 4, size=n)
    k = kind[:, None]
    return np.where(k == 1, ins, np.where(k == 2, dele, c)).astype(np.uint8)


def indel_overlap_pairs(n, L=150, seed=1, gap_limit=5):
    """pairs on which OverlapAnalysis::analyze's one-gap pass (overlapanalysis.cpp:91-139) ACCEPTS.
    Matcher::diffWithOneInsertion (matcher.cpp:56-100) returns -1 as soon as the UNGAPPED mismatch count of a
    prefix plus the last shifted position exceeds the limit - also after a good split was seen - so the pass only
    accepts where the ungapped count over the first overlap_len-2 positions is within the limit while the
    ungapped scan (first min(overlap_len, 50) positions, :34-44) was over it: short overlaps (<= 51) whose last two
    positions mismatch.  Built here: overlap lengths around 30..51 on both sides (read-through = negative offset,
    where the accepted result trims adapters, and barely-overlapping = positive offset), k substitutions in the body
    of the overlap with k around gap_limit, 0..2 substitutions on the last two overlap positions, in either mate;
    plus single-base insertions / deletions and clean pairs as controls."""
    rng = np.random.default_rng(seed)
    j = np.arange(L)[None, :]
    ol = rng.integers(20, 60, size=n)
    ins = np.where(rng.random(n) < 0.6, ol, 2 * L - ol)     # fragment length: overlap = min(ins, L) - max(0, ins - L)
    F = rng.integers(0, 4, size=(n, 2 * L + 1), dtype=np.uint8)
    rnd1 = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    rnd2 = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
    c1 = np.where(j < ins[:, None], np.take_along_axis(F, np.minimum(j, 2 * L) + np.zeros((n, 1), dtype=np.int64), axis=1), rnd1)
    i2 = np.clip(ins[:, None] - 1 - j, 0, 2 * L)
    c2 = np.where(j < ins[:, None], _COMP_CODE[np.take_along_axis(F, i2, axis=1)], rnd2)
    # the overlap is r1[a, a+ol) against rc(r2[a, a+ol)): overlap position t sits at r1[a+t] and at r2[a+ol-1-t]
    a = np.maximum(0, ins - L)
    t = j - a[:, None]                                      # overlap coordinate of r1[j]
    k = np.clip(gap_limit + rng.integers(-3, 3, size=n), 0, None)
    body = (t >= 0) & (t < (ol - 2)[:, None])
    score = np.where(body, rng.random((n, L)), 2.0)
    kth = np.sort(score, axis=1)[np.arange(n), np.minimum(k, L - 1)]
    sub_body = body & (score < kth[:, None]) & (k[:, None] > 0)
    tail = ((t == (ol - 2)[:, None]) & (rng.random((n, 1)) < 0.75)) | ((t == (ol - 1)[:, None]) & (rng.random((n, 1)) < 0.75))
    sub = sub_body | tail                                   # in overlap coordinates, stored on r1's row index
    bump = rng.integers(1, 4, size=(n, L)).astype(np.uint8)  # a substitution always changes the base
    in_r1 = rng.random((n, L)) < 0.5
    c1 = np.where(sub & in_r1, (c1 + bump) & 3, c1).astype(np.uint8)
    # the same overlap positions on r2: r2 index a + ol - 1 - t
    j2 = a[:, None] + ol[:, None] - 1 - t
    sub2 = np.zeros((n, L), dtype=bool)
    rows = np.repeat(np.arange(n)[:, None], L, axis=1)
    ok = sub & ~in_r1 & (j2 >= 0) & (j2 < L)
    sub2[rows[ok], j2[ok]] = True
    c2 = np.where(sub2, (c2 + bump) & 3, c2).astype(np.uint8)
    # controls: 10 % carry a real single-base indel in the body instead
    ind = rng.random(n) < 0.10
    kind = np.where(ind, rng.integers(1, 3, size=n), 0)
    mate = rng.integers(0, 2, size=n)
    ti = (rng.random(n) * np.maximum(1, ol - 4)).astype(np.int64) + 2
    c1 = _indel_rows(c1, np.where(mate == 0, kind, 0), np.clip(a + ti, 0, L - 2), rng)
    c2 = _indel_rows(c2, np.where(mate == 1, kind, 0), np.clip(a + ol - 1 - ti, 0, L - 2), rng)
    outs = {}
    for tag, c in (("1", c1), ("2", c2)):
        lens = np.where(rng.random(n) < 0.9, L, rng.integers(L - 20, L + 1, size=n)).astype(np.int32)
        q = np.where(rng.random((n, L)) < 0.05, rng.integers(2, 15, size=(n, L)), rng.integers(30, 41, size=(n, L)))
        outs["seq" + tag], outs["qual" + tag], outs["len" + tag] = _finish(c, lens, q, L)
    return outs


def adapter_indel_reads(n, L=150, seed=1, paired=True, adapters=None):
    """reads that BEGIN with the adapter carrying one inserted / one deleted base: AdapterTrimmer::trimBySequence's
    one-gap loops compare the read from position 0 whatever `pos` is (adaptertrimmer.cpp:105-135, quirk #7), so
    these are the inputs on which Matcher::matchWithOneInsertion accepts.  The adapter prefix length varies
    (the compare length shrinks with pos, so a short clean prefix matches late in the loop), some reads are
    shorter than the adapter, and controls carry the exact adapter at 0 / at -1..-4 / further inside."""
    rng = np.random.default_rng(seed)
    j = np.arange(L)[None, :]
    outs = {}
    ad1, ad2 = adapters if adapters else (ADAPTER_R1, ADAPTER_R2)
    for tag, adapter in (("1", ad1), ("2", ad2)) if paired else (("1", ad1),):
        if isinstance(adapter, str):
            adapter = adapter.encode()
        acode = np.array([b"ACGT".index(ch) for ch in adapter], dtype=np.uint8)
        alen = len(acode)
        c = rng.integers(0, 4, size=(n, L), dtype=np.uint8)
        what = rng.random(n)
        m = rng.integers(10, alen + 1, size=n)                     # adapter bases placed at the read start
        apad = np.concatenate([acode, rng.integers(0, 4, size=L, dtype=np.uint8)])
        head = (what < 0.85)
        c = np.where(head[:, None] & (j < m[:, None]), apad[np.minimum(j, len(apad) - 1)], c)
        kind = np.where(what < 0.35, 1, np.where(what < 0.7, 2, 0))  # 35 % insertion, 35 % deletion, 15 % exact, rest random
        pos = 2 + (rng.random(n) * np.maximum(1, m - 5)).astype(np.int64)
        c = _indel_rows(c, kind, pos, rng)
        # a few exact adapters shifted left by 1..4 (the Hamming scan's negative start) or placed inside the read
        shl = (what >= 0.7) & (what < 0.76)
        sh = rng.integers(1, 5, size=n)
        c = np.where(shl[:, None], np.take_along_axis(c, np.minimum(j + sh[:, None], L - 1), axis=1), c)
        inner = what >= 0.9
        at = rng.integers(20, L - 10, size=n)
        c = np.where(inner[:, None] & (j >= at[:, None]), apad[np.clip(j - at[:, None], 0, len(apad) - 1)], c)
        e = rng.random((n, L)) < 0.01 * rng.random((n, 1))
        c = np.where(e, rng.integers(0, 4, size=(n, L)), c).astype(np.uint8)
        r = rng.random(n)
        lens = np.where(r < 0.6, L, np.where(r < 0.8, rng.integers(0, alen + 8, size=n), rng.integers(0, L + 1, size=n))).astype(np.int32)
        q = np.where(rng.random((n, L)) < 0.05, rng.integers(2, 15, size=(n, L)), rng.integers(30, 41, size=(n, L)))
        outs["seq" + tag], outs["qual" + tag], outs["len" + tag] = _finish(c, lens, q, L)
    return outs


def to_fastq(seq, qual, lens, mate, name_prefix="@SIM:1:FC:1:1101"):
    """FASTQ bytes; names are Illumina-like but do NOT start with a 2-colour prefix
    (@A/@NS/@NB/@VH/@LH), so the reference's polyG auto-enable stays off."""
    n = len(lens)
    parts = []
    for i in range(n):
        L = int(lens[i])
        parts.append(b"%s:%d:%d %d:N:0:ATCG\n" % (name_prefix.encode(), i // 1000, i, mate))
        parts.append(seq[i, :L].tobytes())
        parts.append(b"\n+\n")
        parts.append(qual[i, :L].tobytes())
        parts.append(b"\n")
    return b"".join(parts)
