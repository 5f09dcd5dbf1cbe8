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
                lowq_site_rate=0.03, paired=True):
    """returns dict(seq1,qual1,len1[,seq2,qual2,len2]) as ASCII uint8 [n, stride] + int32 lens"""
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
