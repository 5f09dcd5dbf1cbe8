"""Host side of the boundary, in Python: what a patched fastp worker does around
one call of the engine (reference: PairEndProcessor::processPairEnd
src/peprocessor.cpp:362-708, SingleEndProcessor::processSingleEnd
src/seprocessor.cpp:197-325).

The engine (GPU library or, in tests, the CPU oracle) returns per-read records;
this module
  * applies them to the original records (prefix/suffix removal + corrections),
  * replays FilterResult::addAdapterTrimmed in input order to maintain the
    adapter-string maps with the reference's caps (filterresult.cpp:7-8,124-180),
  * routes records to the output streams exactly as peprocessor.cpp:563-621 /
    seprocessor.cpp:280-290.
It never computes a trimming decision itself.
"""
from __future__ import annotations

import numpy as np

from . import abi

FAILED_TYPES = [""] * 32  # src/common.h:57-66
FAILED_TYPES[0] = "passed"
FAILED_TYPES[4] = "failed_polyx_filter"
FAILED_TYPES[8] = "failed_bad_overlap"
FAILED_TYPES[12] = "failed_too_many_n_bases"
FAILED_TYPES[16] = "failed_too_short"
FAILED_TYPES[17] = "failed_too_long"
FAILED_TYPES[20] = "failed_quality_filter"
FAILED_TYPES[24] = "failed_low_complexity"
FAILED_TYPES[28] = "failed_adapter_dimer"

MAX_ADAPTER_REC = 20000       # filterresult.cpp:7
LOW_COMPLEXITY_SKIP = 5000    # filterresult.cpp:8

_COMP = bytes.maketrans(b"ACGTacgt", b"TGCATGCA")


def _complement_bytes(b: bytes) -> bytes:
    # util.h:16-33: anything outside ACGTacgt -> 'N'
    out = bytearray(b.translate(_COMP))
    for i, c in enumerate(b):
        if c not in b"ACGTacgt":
            out[i] = ord("N")
    return bytes(out)


class FastqBatch:
    """A ReadPack (src/read.h:66-69) decoded into arrays: names/strands as bytes
    objects, sequence/quality as ASCII rows of a fixed stride."""

    def __init__(self, names, strands, seq, qual, lens):
        self.names = names
        self.strands = strands
        self.seq = seq      # uint8 [n, stride]
        self.qual = qual    # uint8 [n, stride]
        self.lens = lens    # int32 [n]

    @property
    def n(self):
        return len(self.names)

    def slice(self, a, b):
        return FastqBatch(self.names[a:b], self.strands[a:b], self.seq[a:b], self.qual[a:b], self.lens[a:b])


def parse_fastq(data: bytes, stride: int | None = None) -> FastqBatch:
    """Minimal FASTQ decode (4 lines per record, like FastqReader::read
    src/fastqreader.cpp:309-368).  Host code - stays on the CPU by design."""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    n = len(lines) // 4
    names = [lines[4 * i].rstrip(b"\r") for i in range(n)]
    seqs = [lines[4 * i + 1].rstrip(b"\r") for i in range(n)]
    strands = [lines[4 * i + 2].rstrip(b"\r") for i in range(n)]
    quals = [lines[4 * i + 3].rstrip(b"\r") for i in range(n)]
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int32, count=n)
    maxlen = int(lens.max()) if n else 0
    if stride is None:
        stride = max(8, (maxlen + 7) // 8 * 8)
    seq = np.zeros((n, stride), dtype=np.uint8)
    qual = np.zeros((n, stride), dtype=np.uint8)
    for i in range(n):
        L = lens[i]
        if L:
            seq[i, :L] = np.frombuffer(seqs[i], dtype=np.uint8)
            qual[i, :L] = np.frombuffer(quals[i][:L].ljust(L, b"!"), dtype=np.uint8)
    return FastqBatch(names, strands, seq, qual, lens)


class AdapterMaps:
    """FilterResult::mAdapter1/mAdapter2 with the reference's insertion caps
    (filterresult.cpp:124-180).  Host-side by design: string keyed."""

    def __init__(self):
        self.a1: dict[bytes, int] = {}
        self.a2: dict[bytes, int] = {}

    @staticmethod
    def _low_complexity(a: bytes) -> bool:  # filterresult.cpp:115-122
        diff = sum(1 for i in range(len(a) - 1) if a[i] != a[i + 1])
        return diff < len(a) // 2

    def _add(self, m, adapter):
        if adapter in m:
            m[adapter] += 1
            return True
        if len(m) > MAX_ADAPTER_REC or (len(m) > LOW_COMPLEXITY_SKIP and self._low_complexity(adapter)):
            return False
        m[adapter] = 1
        return True

    def add_single(self, adapter: bytes, is_r2: bool):  # :124-152
        if not adapter:
            return
        self._add(self.a2 if is_r2 else self.a1, adapter)

    def add_pair(self, a1: bytes, a2: bytes):  # :154-180 (early return quirk #8)
        if a1:
            if not self._add(self.a1, a1):
                return
        if a2:
            self._add(self.a2, a2)


class Outputs:
    def __init__(self, paired, want_failed=False, want_unpaired1=False, want_unpaired2=False):
        self.out1 = bytearray()
        self.out2 = bytearray() if paired else None
        self.failed = bytearray() if want_failed else None
        self.unpaired1 = bytearray() if want_unpaired1 else None
        self.unpaired2 = bytearray() if want_unpaired2 else None
        self.merged = bytearray()
        self.overlapped = bytearray()   # --overlapped_out (peprocessor.cpp:488-495)


def _record(name, seq, strand, qual, tag=None):  # Read::appendToString read.cpp:119-154
    if tag is not None:
        name = name + b" " + tag.encode()
    return name + b"\n" + seq + b"\n" + strand + b"\n" + qual + b"\n"


def _final_read(batch: FastqBatch, i: int, rr, corr_for_read):
    """orig[front:front+len] after applying the sparse corrections."""
    L0 = int(batch.lens[i])
    s = bytearray(batch.seq[i, :L0].tobytes())
    q = bytearray(batch.qual[i, :L0].tobytes())
    for pos, base, ql in corr_for_read:
        s[pos] = base
        q[pos] = ql
    return s, q


class UmiNameEditor:
    """The name edit of UmiProcessor::process for UMIs taken from the read itself
    (umiprocessor.cpp:19-61; locations read1 / read2 / per_read).  Host-side by
    design: the engine only reproduces the sequence front-trim (params.umi_len*)."""

    def __init__(self, loc: str, umi_len: int, prefix: bytes = b"", delimiter: bytes = b":"):
        assert loc in ("read1", "read2", "per_read")
        self.loc, self.umi_len, self.prefix, self.delimiter = loc, umi_len, prefix, delimiter

    def _tagged(self, name: bytes, umi: bytes) -> bytes:  # addUmiToName :62-81
        tag = self.delimiter + (self.prefix + b"_" if self.prefix else b"") + umi
        sp = name.find(b" ")
        return name + tag if sp < 0 else name[:sp] + tag + name[sp:]

    def edit(self, name1: bytes, seq1: bytes, name2: bytes | None, seq2: bytes | None):
        if self.loc == "read1":
            umi = seq1[:self.umi_len]
        elif self.loc == "read2":
            if seq2 is None:
                return name1, name2
            umi = seq2[:self.umi_len]
        else:
            umi = seq1[:self.umi_len]
            if seq2 is not None:
                umi = umi + b"_" + seq2[:self.umi_len]
            return self._tagged(name1, umi), (self._tagged(name2, umi) if name2 is not None else None)
        if not umi:
            return name1, name2
        return self._tagged(name1, umi), (self._tagged(name2, umi) if name2 is not None else None)


def apply_results(params: abi.Params, b1: FastqBatch, b2: FastqBatch | None, r1, r2, pair, corrections,
                  outputs: Outputs, amaps: AdapterMaps, umi: UmiNameEditor | None = None, adapter_events=None):
    """Turn engine results for one pack into output records + adapter-map updates."""
    paired = b2 is not None
    if umi is not None:  # names are edited before anything is routed (peprocessor.cpp:419-420)
        names1, names2 = list(b1.names), (list(b2.names) if paired else None)
        for i in range(b1.n):
            o1 = b1.seq[i, :int(b1.lens[i])].tobytes()
            o2 = b2.seq[i, :int(b2.lens[i])].tobytes() if paired else None
            n1, n2 = umi.edit(names1[i], o1, names2[i] if paired else None, o2)
            names1[i] = n1
            if paired:
                names2[i] = n2
        b1 = FastqBatch(names1, b1.strands, b1.seq, b1.qual, b1.lens)
        if paired:
            b2 = FastqBatch(names2, b2.strands, b2.seq, b2.qual, b2.lens)
    corr_by_read: dict[int, list] = {}
    if corrections is not None:
        for c in corrections:
            corr_by_read.setdefault(int(c["read"]), []).append((int(c["pos"]), int(c["base"]), int(c["qual"])))
    a1seq = params.adapter_seq_r1 or b""
    a2seq = params.adapter_seq_r2 or b""
    fasta = abi.adapter_fasta_list(params) if params.n_adapter_fasta else []
    ev_by_read: dict[int, list] = {}
    if adapter_events is not None:
        for e in adapter_events:  # already ordered per read by adapter index
            ev_by_read.setdefault(int(e["read"]), []).append((int(e["pos"]), int(e["len"]), int(e["adapter"])))
    for i in range(b1.n):
        rr1 = r1[i]
        s1, q1 = _final_read(b1, i, rr1, corr_by_read.get(2 * i if paired else i, ()))
        if paired:
            rr2 = r2[i]
            s2, q2 = _final_read(b2, i, rr2, corr_by_read.get(2 * i + 1, ()))

        def adapter_string(rr, s, aseq):
            pos, alen = int(rr["adapter_pos"]), int(rr["adapter_len"])
            if pos < 0:
                return bytes(aseq[:alen])
            f = int(rr["front"])
            return bytes(s[f + pos:f + pos + alen])

        def replay_fasta(read_key, rr, s, is_r2):  # trimByMultiSequences adaptertrimmer.cpp:48-62
            f = int(rr["front"])
            for pos, alen, ai in ev_by_read.get(read_key, ()):
                amaps.add_single(bytes(fasta[ai][:alen]) if pos < 0 else bytes(s[f + pos:f + pos + alen]), is_r2)

        # --- adapter string map, in input order -------------------------------
        if paired:
            ov_trim = (rr1["flags"] & abi.RF_ADAPTER_OV) != 0
            if ov_trim:  # adaptertrimmer.cpp:27-42
                amaps.add_pair(adapter_string(rr1, s1, a1seq), adapter_string(rr2, s2, a2seq))
            else:  # adapter_len == 0: the flag comes from an --adapter_fasta trim only
                if (rr1["flags"] & abi.RF_ADAPTER) and rr1["adapter_len"]:
                    amaps.add_single(adapter_string(rr1, s1, a1seq), False)
                if (rr2["flags"] & abi.RF_ADAPTER) and rr2["adapter_len"]:
                    amaps.add_single(adapter_string(rr2, s2, a2seq), True)
            replay_fasta(2 * i, rr1, s1, False)      # peprocessor.cpp:467-470
            replay_fasta(2 * i + 1, rr2, s2, True)
        else:
            if (rr1["flags"] & abi.RF_ADAPTER) and rr1["adapter_len"]:
                amaps.add_single(adapter_string(rr1, s1, a1seq), False)
            replay_fasta(i, rr1, s1, False)          # seprocessor.cpp:249-251

        def cut(rr, s, q):
            f, L = int(rr["front"]), int(rr["len"])
            return bytes(s[f:f + L]), bytes(q[f:f + L])

        t1s, t1q = cut(rr1, s1, q1)
        dup = (rr1["flags"] & abi.RF_DUP) != 0
        dedup_out = bool(params.dedup) and dup
        code1 = int(rr1["code"])
        alive1 = not (rr1["flags"] & abi.RF_NULL)
        if not paired:  # seprocessor.cpp:280-290
            if not dedup_out:
                if alive1 and code1 == abi.PASS_FILTER:
                    outputs.out1 += _record(b1.names[i], t1s, b1.strands[i], t1q)
                elif outputs.failed is not None:
                    outputs.failed += _record(b1.names[i], t1s, b1.strands[i], t1q, FAILED_TYPES[code1])
            continue
        t2s, t2q = cut(rr2, s2, q2)
        code2 = int(rr2["code"])
        alive2 = not (rr2["flags"] & abi.RF_NULL)
        if params.overlapped_out and (int(rr1["reserved"]) & abi.OVOUT_HIT):  # peprocessor.cpp:488-495
            # string(substr(max(0, offset)), overlap_len) at :491 is std::string's (str, pos) constructor: what is
            # printed are the bases of read 1 BEHIND the region the exact (diffPercentLimit 0) analysis found
            # overlapped, of the read as it was right after adapter trimming (later polyX / max_len cuts do not apply)
            f, st, cnt = int(rr1["front"]), int(rr1["reserved"]) & 0x7FFF, int(rr2["reserved"])
            outputs.overlapped += _record(b1.names[i], bytes(s1[f + st:f + st + cnt]), b1.strands[i], bytes(q1[f + st:f + st + cnt]))
        if params.merge and alive1 and alive2:  # peprocessor.cpp:518-561
            if pair[i]["flags"] & abi.PF_OVERLAPPED:
                if code1 == abi.PASS_FILTER:
                    # OverlapAnalysis::merge overlapanalysis.cpp:148-179
                    ol = int(pair[i]["ov_len"])
                    if params.overlapped_out:   # the reserved fields are taken: the part lengths follow from the pair record
                        off = int(pair[i]["ov_offset"])
                        m1, m2 = ol + max(0, off), (len(t2s) - ol if off > 0 else 0)
                    else:
                        m1, m2 = int(rr1["reserved"]), int(rr2["reserved"])
                    rc2 = _complement_bytes(t2s[::-1])
                    rq2 = t2q[::-1]
                    tag = b" merged_%d_%d" % (m1, m2)
                    strand = b1.strands[i] if b1.strands[i] == b"+" else b1.strands[i] + tag
                    outputs.merged += _record(b1.names[i] + tag, t1s[:m1] + rc2[ol:ol + m2], strand,
                                              t1q[:m1] + rq2[ol:ol + m2])
                continue
            if params.merge_include_unmerged:
                if code1 == abi.PASS_FILTER and not dedup_out:
                    outputs.merged += _record(b1.names[i], t1s, b1.strands[i], t1q)
                if code2 == abi.PASS_FILTER and not dedup_out:
                    outputs.merged += _record(b2.names[i], t2s, b2.strands[i], t2q)
                continue
        if dedup_out:
            continue
        p1 = alive1 and code1 == abi.PASS_FILTER
        p2 = alive2 and code2 == abi.PASS_FILTER
        if p1 and p2:  # peprocessor.cpp:577-594
            outputs.out1 += _record(b1.names[i], t1s, b1.strands[i], t1q)
            outputs.out2 += _record(b2.names[i], t2s, b2.strands[i], t2q)
        elif p1:  # :595-605
            if outputs.unpaired1 is not None:
                outputs.unpaired1 += _record(b1.names[i], t1s, b1.strands[i], t1q)
                if outputs.failed is not None:
                    outputs.failed += _record(b2.names[i], t2s, b2.strands[i], t2q, FAILED_TYPES[code2])
            elif outputs.failed is not None:
                outputs.failed += _record(b1.names[i], t1s, b1.strands[i], t1q, "paired_read_is_failing")
                outputs.failed += _record(b2.names[i], t2s, b2.strands[i], t2q, FAILED_TYPES[code2])
        elif p2:  # :606-621
            if outputs.unpaired2 is not None:
                outputs.unpaired2 += _record(b2.names[i], t2s, b2.strands[i], t2q)
                if outputs.failed is not None:
                    outputs.failed += _record(b1.names[i], t1s, b1.strands[i], t1q, FAILED_TYPES[code1])
            elif outputs.unpaired1 is not None:
                outputs.unpaired1 += _record(b2.names[i], t2s, b2.strands[i], t2q)
                if outputs.failed is not None:
                    outputs.failed += _record(b1.names[i], t1s, b1.strands[i], t1q, FAILED_TYPES[code1])
            elif outputs.failed is not None:
                outputs.failed += _record(b1.names[i], t1s, b1.strands[i], t1q, FAILED_TYPES[code1])
                outputs.failed += _record(b2.names[i], t2s, b2.strands[i], t2q, "paired_read_is_failing")
