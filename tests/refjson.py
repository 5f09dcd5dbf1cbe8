"""Rebuild fastp's JSON report numbers from an engine counter block.

Test infrastructure: in a real drop-in the counter block is loaded back into the
reference's own Stats/FilterResult objects and *its* JsonReporter prints the
report (jsonreporter.cpp:22-171).  Here we recompute the same numbers so the
parity tests can compare them with the JSON the reference binary wrote.
Every double is rounded through '%g' (precision 6), which is what the
reference's `ofs << double` emits, so equality below == textual equality.
"""
import json

import numpy as np

from fastp_amd import abi

BIN = {"A": ord("A") & 7, "T": ord("T") & 7, "C": ord("C") & 7, "G": ord("G") & 7, "N": ord("N") & 7}


def g6(x):
    return float("%g" % x)


def _div(a, b):
    return a / b


class StatsView:
    """Stats::summarize (stats.cpp:102-182) over one Stats slot of the counter block."""

    def __init__(self, ctr, lay, which, params=None):
        base = lay.stats[which]
        C = int(lay.cycles)
        # overrepresentation analysis: mOverRepSeq in map (= seed list) order
        self.overrep = {}
        n = int(lay.n_overrep[which])
        if params is not None and params.overrep_enabled and n:
            seeds = abi.overrep_lists(params)[1 if which >= 2 else 0]
            cnt = ctr[lay.overrep_count[which]: lay.overrep_count[which] + n]
            s = int(params.overrep_sampling)
            thr = {10: 500, 20: 200, 40: 100, 100: 50}          # Stats::overRepPassed stats.cpp:519-533
            for seq, c in zip(seeds, cnt):
                if s * int(c) > thr.get(len(seq), 20):
                    self.overrep[seq.decode()] = int(c)
        self.reads = int(ctr[base + lay.st_reads])
        self.length_sum = int(ctr[base + lay.st_length_sum])
        self.qual_hist = ctr[base + lay.st_qual_hist: base + lay.st_qual_hist + 128]
        self.kmer = ctr[base + lay.st_kmer: base + lay.st_kmer + 1024]
        cyc = ctr[base + lay.st_cycle: base + lay.st_cycle + 34 * C].reshape(34, C)
        self.q30 = cyc[0:8]
        self.q20 = cyc[8:16]
        self.content = cyc[16:24]
        self.qual = cyc[24:32]
        self.total_base = cyc[32]
        self.total_qual = cyc[33]
        zeros = np.nonzero(self.total_base == 0)[0]
        self.cycles = int(zeros[0]) if len(zeros) else C
        c = self.cycles
        self.bases = int(self.total_base[:c].sum())
        self.q20_total = int(self.q20[:, :c].sum())
        self.q30_total = int(self.q30[:, :c].sum())
        self.q40_total = int(self.qual_hist[73:127].sum())
        self.gc = int(self.content[BIN["G"], :c].sum() + self.content[BIN["C"], :c].sum())

    def mean_length(self):  # stats.cpp:184-189 (integer division)
        return 0 if self.reads == 0 else self.length_sum // self.reads

    def report(self):  # Stats::reportJson stats.cpp:374-463
        c = self.cycles
        tb = self.total_base[:c].astype(np.float64)
        mean = self.total_qual[:c] / tb if c else np.zeros(0)
        qc = {}
        cc = {}
        for name in "ATCG":
            b = BIN[name]
            cont = self.content[b, :c].astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                q = np.where(cont == 0, mean, self.qual[b, :c] / cont)
            qc[name] = [g6(v) for v in q]
            cc[name] = [g6(v) for v in cont / tb]
        qc["mean"] = [g6(v) for v in mean]
        cc["N"] = [g6(v) for v in self.content[BIN["N"], :c] / tb]
        cc["GC"] = [g6(v) for v in (self.content[BIN["G"], :c] + self.content[BIN["C"], :c]) / tb]
        letters = "ATCG"
        kmer = {}
        for i in range(1024):
            key = "".join(letters[(i >> s) & 3] for s in (8, 6, 4, 2, 0))
            kmer[key] = int(self.kmer[i])
        return {
            "total_reads": self.reads, "total_bases": self.bases, "q20_bases": self.q20_total,
            "q30_bases": self.q30_total, "q40_bases": self.q40_total, "total_cycles": c,
            "quality_curves": qc, "content_curves": cc, "kmer_count": kmer,
            "overrepresented_sequences": self.overrep,
        }


def _adapter_counts(m):  # FilterResult::outputAdaptersJson filterresult.cpp:255-292
    total = sum(m.values())
    if total == 0:
        return {}
    out = {}
    reported = 0
    for k in sorted(m, key=lambda s: (len(s), s)):  # classcomp filterresult.h:14-23
        v = m[k]
        if v / float(total) < 0.01:
            continue
        out[k.decode()] = v
        reported += v
    if total - reported > 0:
        out["others"] = total - reported
    return out


def build(ctr, lay, params: abi.Params, amaps=None):
    """dict with the same keys/values as the reference JSON (minus 'command')."""
    paired = bool(params.paired)
    pre1 = StatsView(ctr, lay, abi.STATS_PRE1, params)
    post1 = StatsView(ctr, lay, abi.STATS_POST1, params)
    pre2 = StatsView(ctr, lay, abi.STATS_PRE2, params) if paired else None
    post2 = StatsView(ctr, lay, abi.STATS_POST2, params) if paired else None

    def summ(s1, s2, with_r2_len):
        reads = s1.reads + (s2.reads if s2 else 0)
        bases = s1.bases + (s2.bases if s2 else 0)
        q20 = s1.q20_total + (s2.q20_total if s2 else 0)
        q30 = s1.q30_total + (s2.q30_total if s2 else 0)
        gc = s1.gc + (s2.gc if s2 else 0)
        d = {"total_reads": reads, "total_bases": bases, "q20_bases": q20, "q30_bases": q30,
             "q20_rate": g6(0.0 if bases == 0 else q20 / bases),
             "q30_rate": g6(0.0 if bases == 0 else q30 / bases),
             "read1_mean_length": s1.mean_length()}
        if with_r2_len:
            d["read2_mean_length"] = s2.mean_length()
        d["gc_content"] = g6(0.0 if bases == 0 else gc / bases)
        return d

    if paired:
        seqinfo = "paired end (%d cycles + %d cycles)" % (pre1.cycles, pre2.cycles)
    else:
        seqinfo = "single end (%d cycles)" % pre1.cycles
    out = {"summary": {"fastp_version": "1.3.6", "sequencing": seqinfo,
                       "before_filtering": summ(pre1, pre2, paired),
                       "after_filtering": summ(post1, post2, paired and not params.merge)}}
    fs = ctr[lay.filter_stats: lay.filter_stats + 32]
    fr = {"passed_filter_reads": int(fs[abi.PASS_FILTER])}
    if params.correction:
        fr["corrected_reads"] = int(ctr[lay.corrected_reads])
        fr["corrected_bases"] = int(ctr[lay.correction: lay.correction + 64].sum())
    fr["low_quality_reads"] = int(fs[abi.FAIL_QUALITY])
    fr["too_many_N_reads"] = int(fs[abi.FAIL_N_BASE])
    if params.complexity_filter:
        fr["low_complexity_reads"] = int(fs[abi.FAIL_COMPLEXITY])
    if params.adapter_enabled:
        fr["adapter_dimer_reads"] = int(fs[abi.FAIL_ADAPTER_DIMER])
    fr["too_short_reads"] = int(fs[abi.FAIL_LENGTH])
    fr["too_long_reads"] = int(fs[abi.FAIL_TOO_LONG])
    out["filtering_result"] = fr
    if params.dup_enabled:
        tot = int(ctr[lay.dup_total])
        out["duplication"] = {"rate": g6(0.0 if tot == 0 else int(ctr[lay.dup_count]) / tot)}
    if paired:
        M = params.insert_size_max
        hist = ctr[lay.isize: lay.isize + M + 1]
        peak, mx = 0, -1
        for i in range(M):  # getPeakInsertSize peprocessor.cpp:338-348
            if hist[i] > mx:
                peak, mx = i, hist[i]
        out["insert_size"] = {"peak": peak, "unknown": int(hist[M]), "histogram": [int(v) for v in hist[:M]]}
    if params.adapter_enabled:
        ac = {"adapter_trimmed_reads": int(ctr[lay.adapter_reads]),
              "adapter_trimmed_bases": int(ctr[lay.adapter_bases])}
        if amaps is not None:
            ac["read1_adapter_counts"] = _adapter_counts(amaps.a1)
            if paired:
                ac["read2_adapter_counts"] = _adapter_counts(amaps.a2)
        out["adapter_cutting"] = ac
    if params.poly_x:
        names = "ATCG"
        pr_ = ctr[lay.polyx_reads: lay.polyx_reads + 4]
        pb = ctr[lay.polyx_bases: lay.polyx_bases + 4]
        out["polyx_trimming"] = {
            "total_polyx_trimmed_reads": int(pr_.sum()),
            "polyx_trimmed_reads": {names[b]: int(pr_[b]) for b in range(4)},
            "total_polyx_trimmed_bases": int(pb.sum()),
            "polyx_trimmed_bases": {names[b]: int(pb[b]) for b in range(4)}}
    out["read1_before_filtering"] = pre1.report()
    if paired:
        out["read2_before_filtering"] = pre2.report()
    out["merged_and_filtered" if params.merge else "read1_after_filtering"] = post1.report()
    if paired and not params.merge:
        out["read2_after_filtering"] = post2.report()
    return out


def load_reference_json(path):
    with open(path) as f:
        d = json.load(f)
    d.pop("command", None)
    return d


def diff(ref, mine, path="", out=None, limit=40, skip=("read1_adapter_sequence", "read2_adapter_sequence")):
    """list of human-readable differences between two report dicts"""
    if out is None:
        out = []
    if len(out) >= limit:
        return out
    if isinstance(ref, dict) and isinstance(mine, dict):
        for k in ref:
            if k in skip:
                continue
            if k not in mine:
                out.append(f"{path}/{k}: missing in engine report")
            else:
                diff(ref[k], mine[k], f"{path}/{k}", out, limit, skip)
        for k in mine:
            if k not in ref:
                out.append(f"{path}/{k}: not in reference report")
    elif isinstance(ref, list) and isinstance(mine, list):
        if len(ref) != len(mine):
            out.append(f"{path}: length {len(ref)} vs {len(mine)}")
        else:
            for i, (a, b) in enumerate(zip(ref, mine)):
                if a != b:
                    out.append(f"{path}[{i}]: ref {a!r} engine {b!r}")
                    if len(out) >= limit:
                        break
    else:
        if ref != mine:
            out.append(f"{path}: ref {ref!r} engine {mine!r}")
    return out
