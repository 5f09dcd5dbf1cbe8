"""helpers for the device FASTQ formatter tests: text -> fastp_gpu_parse_fastq -> fastp_gpu_submit_device ->
fastp_gpu_format_fastq, all on 'device' buffers (hostsim: numpy arrays; GPU: torch tensors)"""
import ctypes as C

import numpy as np

from fastp_amd import abi


class NumpyMem:
    """hostsim: device pointers are host pointers"""

    def __init__(self):
        self.keep = []

    def alloc(self, nbytes, fill=0):
        raw = np.full(nbytes + 16, fill, dtype=np.uint8)
        o = (-raw.ctypes.data) % 16
        a = raw[o:o + nbytes]
        self.keep.append(raw)
        return a

    def upload(self, data: bytes, pad=0):
        a = self.alloc(len(data) + pad)
        a[:len(data)] = np.frombuffer(data, dtype=np.uint8)
        return a

    def ptr(self, a):
        return a.ctypes.data

    def download(self, a, n=None):
        return bytes(a[:n] if n is not None else a)

    def sync(self):
        pass


class TorchMem:
    def __init__(self, device=0):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device)

    def alloc(self, nbytes, fill=0):
        return self.torch.full((max(16, nbytes),), fill, dtype=self.torch.uint8, device=self.dev)

    def upload(self, data: bytes, pad=0):
        t = self.torch.frombuffer(bytearray(data + b"\0" * pad), dtype=self.torch.uint8).to(self.dev)
        assert t.data_ptr() % 16 == 0
        return t

    def ptr(self, a):
        return a.data_ptr()

    def download(self, a, n=None):
        return (a[:n] if n is not None else a).cpu().numpy().tobytes()

    def sync(self):
        self.torch.cuda.synchronize(self.dev)


def _prepare(eng, mem, fq1: bytes, fq2, max_len, corr_cap=1 << 16):
    """text -> parse -> submit_device; everything stays in `mem`"""
    paired = fq2 is not None
    ss, qs = abi.seq_stride(max_len), abi.qual_stride(max_len)
    mates = []
    n = None
    for txt in ((fq1, fq2) if paired else (fq1,)):
        cap = txt.count(b"\n") // 4 + 2
        t = mem.upload(txt, (-len(txt)) % 16 + 16)
        seq, qual = mem.alloc(cap * ss), mem.alloc(cap * qs)
        lens, loff, llen = mem.alloc(cap * 2), mem.alloc(cap * 16), mem.alloc(cap * 16)
        mem.sync()
        info = eng.parse_fastq(mem.ptr(t), len(txt), True, cap, mem.ptr(seq), mem.ptr(qual), mem.ptr(lens),
                               mem.ptr(loff), mem.ptr(llen))
        assert info.first_bad == -1
        n = info.n_records if n is None else min(n, info.n_records)
        mates.append(dict(text=t, seq=seq, qual=qual, lens=lens, loff=loff, llen=llen, nbytes=len(txt)))
    res = [mem.alloc(n * 12), mem.alloc(n * 12)]
    pr = mem.alloc(n * 8)
    corr = mem.alloc(corr_cap * 8)
    nc = mem.alloc(16)
    b = abi.Batch()
    b.n, b.flags = n, abi.BATCH_STAT_ISIZE
    b.seq1, b.qual1, b.len1 = (mem.ptr(mates[0][k]) for k in ("seq", "qual", "lens"))
    if paired:
        b.seq2, b.qual2, b.len2 = (mem.ptr(mates[1][k]) for k in ("seq", "qual", "lens"))
    r = abi.Results()
    r.r1, r.r2, r.pair = mem.ptr(res[0]), mem.ptr(res[1]) if paired else None, mem.ptr(pr) if paired else None
    r.corrections, r.corrections_capacity, r.n_corrections = mem.ptr(corr), corr_cap, mem.ptr(nc)
    ev_cap = 4 * n + 16
    ev, nev = mem.alloc(ev_cap * 12), mem.alloc(16)
    r.adapter_events, r.adapter_events_capacity, r.n_adapter_events = mem.ptr(ev), ev_cap, mem.ptr(nev)
    mem.sync()
    eng.submit_device(b, r)
    eng.synchronize()
    return dict(n=n, mates=mates, res=res, pair=pr, corr=corr, nc=nc, paired=paired)


def run(eng, mem, params, fq1: bytes, fq2, max_len, out_slack=0, corr_cap=1 << 16):
    """returns (rc, out1 bytes, out2 bytes | None, (len1, len2)) of the whole text as ONE batch"""
    c = _prepare(eng, mem, fq1, fq2, max_len, corr_cap)
    n, mates, res, corr, nc, paired = c["n"], c["mates"], c["res"], c["corr"], c["nc"], c["paired"]
    fin = []
    for m in range(2 if paired else 1):
        f = abi.FormatIn()
        f.text, f.line_off, f.line_len, f.res = (mem.ptr(mates[m]["text"]), mem.ptr(mates[m]["loff"]),
                                                 mem.ptr(mates[m]["llen"]), mem.ptr(res[m]))
        fin.append(f)
    caps = [mates[m]["nbytes"] + 4 + out_slack for m in range(len(mates))]
    outs = [mem.alloc(max(16, c), 0xEE) for c in caps]
    mem.sync()
    rc, l1, l2 = eng.format_fastq(n, fin[0], fin[1] if paired else None, mem.ptr(corr), mem.ptr(nc), mem.ptr(outs[0]),
                                  caps[0], mem.ptr(outs[1]) if paired else None, caps[1] if paired else 0, check=False)
    o1 = mem.download(outs[0], min(l1, caps[0]))
    o2 = mem.download(outs[1], min(l2, caps[1])) if paired else None
    return rc, o1, o2, (l1, l2)


STREAMS = ("out1", "out2", "failed", "merged", "unpaired1", "unpaired2")


def run_streams(eng, mem, params, fq1: bytes, fq2, max_len, want_failed=True, want_unpaired=False, umi=None,
                shrink=None):
    """every output stream through fastp_gpu_format_streams; umi = (loc, len[, prefix, delimiter]) or None.
    returns (rc, {stream: bytes}, [needed lengths]); shrink = stream index whose buffer is made too small"""
    c = _prepare(eng, mem, fq1, fq2, max_len)
    n, mates, res, paired = c["n"], c["mates"], c["res"], c["paired"]
    ios = []
    for m in range(2 if paired else 1):
        f = abi.FormatIn()
        f.text, f.line_off, f.line_len, f.res = (mem.ptr(mates[m]["text"]), mem.ptr(mates[m]["loff"]),
                                                 mem.ptr(mates[m]["llen"]), mem.ptr(res[m]))
        ios.append(f)
    o = abi.FormatOptions()
    o.want_failed, o.want_unpaired1, o.want_unpaired2 = int(want_failed), int(want_unpaired), int(want_unpaired)
    if umi is not None:
        o.umi_loc = {"read1": 1, "read2": 2, "per_read": 3}[umi[0]]
        o.umi_len = umi[1]
        o.umi_prefix = umi[2] if len(umi) > 2 and umi[2] else None
        o.umi_delimiter = umi[3] if len(umi) > 3 else None
    total = sum(m["nbytes"] for m in mates)
    # worst case of one stream: every record of both mates, each with a UMI tag, a merged tag and a failed tag
    cap = total + n * 2 * 160 + 64
    caps = [cap] * 6
    if shrink is not None:
        caps[shrink] = 100
    outs = [mem.alloc(max(16, k), 0xEE) for k in caps]
    mem.sync()
    rc, lens = eng.format_streams(n, ios[0], ios[1] if paired else None, mem.ptr(c["pair"]) if paired else None,
                                  mem.ptr(c["corr"]), mem.ptr(c["nc"]), o, [mem.ptr(x) for x in outs], caps, check=False)
    got = {k: mem.download(outs[i], min(lens[i], caps[i])) for i, k in enumerate(STREAMS)}
    for i in range(6):  # nothing past the reported length
        tail = mem.download(outs[i])[min(lens[i], caps[i]):]
        assert tail.count(b"\xEE") == len(tail), f"stream {STREAMS[i]}: bytes written past its length"
    return rc, got, lens
