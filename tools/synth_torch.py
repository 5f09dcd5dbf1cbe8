"""Synthetic 2xL paired-end batches generated directly in HBM with torch ops (bench / GPU tests).

Same fragment model as tests/synth.py (SURVEY.md 8d "S-B"): insert ~ N(mean, sd), TruSeq
read-through + poly-A pad when the insert is shorter than the read, Q37 plateau with a drop
point, substitution errors at 10^(-Q/10), low-quality sites turning into N, optional exact
duplicates.  Not bit-identical to the numpy generator (different RNG) - parity is always
judged against the oracle / reference run on the SAME generated reads.

Returns ASCII tensors (uint8 [n, L]) and packs them into the engine's SoA layout
(include/fastp_gpu.h) with torch bit ops; all of this is data preparation, outside any
timed region.
"""
import torch

ADAPTER_R1 = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
ADAPTER_R2 = b"AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"


def _hash64(x):
    # splitmix-style integer hash on int64 tensors (wraparound arithmetic)
    x = (x ^ (x >> 30)) * -4658895280553007687   # 0xBF58476D1CE4E5B9
    x = (x ^ (x >> 27)) * -7723592293110705685   # 0x94D049BB133111EB
    return x ^ (x >> 31)


def synth_pairs_torch(n, L=150, seed=42, device="cuda", insert_mean=300.0, insert_sd=80.0, insert_min=20,
                      insert_max=800, dup_frac=0.10, n_rate=0.5, lowq_site_rate=0.03, chunk=1 << 20):
    """dict(seq1, qual1, len1, seq2, qual2, len2): uint8 [n, L] ASCII tensors + int32 lens (all == L)"""
    outs = {k: [] for k in ("seq1", "qual1", "seq2", "qual2")}
    acgt = torch.tensor(list(b"ATCG"), dtype=torch.uint8, device=device)   # code order of the engine
    comp = torch.tensor([1, 0, 3, 2], dtype=torch.int64, device=device)
    pads = []
    for ad in (ADAPTER_R1, ADAPTER_R2):
        pads.append(torch.tensor(list(ad) + [ord("A")] * (L + 8), dtype=torch.uint8, device=device))
    for start in range(0, n, chunk):
        m = min(chunk, n - start)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1000003 + start)
        frag_id = torch.arange(start, start + m, device=device, dtype=torch.int64)
        if dup_frac > 0:
            # a duplicate re-uses the fragment identity (hence bases and insert size) of an earlier pair
            is_dup = torch.rand(m, generator=g, device=device) < dup_frac
            src = (torch.rand(m, generator=g, device=device).double() * frag_id.double()).long()
            frag_id = torch.where(is_dup & (frag_id > 0), src, frag_id)
        # make the insert size a deterministic function of frag_id (so duplicates agree)
        u1 = ((_hash64(frag_id * 2 + 1 + seed * 1315423911) >> 11) & ((1 << 52) - 1)).double() / float(1 << 52)
        u2 = ((_hash64(frag_id * 2 + 2 + seed * 1315423911) >> 11) & ((1 << 52) - 1)).double() / float(1 << 52)
        z = torch.sqrt(-2.0 * torch.log(u1.clamp_min(1e-300))) * torch.cos(6.283185307179586 * u2)
        ins = (insert_mean + insert_sd * z).round().clamp(insert_min, insert_max).long()
        j = torch.arange(L, device=device, dtype=torch.int64)[None, :]
        for mate, pad in ((1, pads[0]), (2, pads[1])):
            inside = j < ins[:, None]
            t = j if mate == 1 else (ins[:, None] - 1 - j)
            code = (_hash64(frag_id[:, None] * 1024 + t.clamp_min(0) + seed * 2654435761) >> 13) & 3
            if mate == 2:
                code = comp[code]
            s = torch.where(inside, acgt[code], pad[(j - ins[:, None]).clamp(0, pad.numel() - 1)])
            drop = torch.randint(L // 2, L + 40, (m, 1), generator=g, device=device)
            lowq = torch.randint(2, 25, (m, 1), generator=g, device=device)
            q = torch.where(j < drop, torch.full_like(lowq, 37), lowq).expand(m, L).clone()
            site = torch.rand(m, L, generator=g, device=device) < lowq_site_rate
            q = torch.where(site, torch.randint(2, 30, (m, L), generator=g, device=device), q)
            perr = torch.pow(10.0, -q.float() / 10.0)
            e = torch.rand(m, L, generator=g, device=device) < perr
            sub = acgt[torch.randint(0, 4, (m, L), generator=g, device=device)]
            s = torch.where(e, sub, s)
            isn = (q <= 2) & (torch.rand(m, L, generator=g, device=device) < n_rate)
            s = torch.where(isn, torch.full_like(s, ord("N")), s)
            outs[f"seq{mate}"].append(s.to(torch.uint8))
            outs[f"qual{mate}"].append((q + 33).to(torch.uint8))
    res = {k: torch.cat(v, 0) for k, v in outs.items()}
    res["len1"] = torch.full((n,), L, dtype=torch.int32, device=device)
    res["len2"] = torch.full((n,), L, dtype=torch.int32, device=device)
    return res


def pack_torch(seq, qual, lens, max_len):
    """ASCII [n, L] -> (seq2 [n, seq_stride] u8, qual [n, qual_stride] u8, len [n] u16-as-int16 view)"""
    n, L = seq.shape
    dev = seq.device
    ss = ((max_len + 3) // 4 + 7) // 8 * 8
    qs = (max_len + 7) // 8 * 8
    lut = torch.zeros(256, dtype=torch.uint8, device=dev)
    for ch, c in zip(b"ATCG", range(4)):
        lut[ch] = c
    valid = torch.arange(L, device=dev)[None, :] < lens[:, None]
    code = torch.where(valid, lut[seq.long()], torch.zeros_like(seq))
    Lp = (L + 3) // 4 * 4
    if Lp != L:
        code = torch.nn.functional.pad(code, (0, Lp - L))
    c4 = code.view(n, Lp // 4, 4).to(torch.int32)
    packed = (c4[:, :, 0] | (c4[:, :, 1] << 2) | (c4[:, :, 2] << 4) | (c4[:, :, 3] << 6)).to(torch.uint8)
    so = torch.zeros(n, ss, dtype=torch.uint8, device=dev)
    so[:, :Lp // 4] = packed
    isn = (seq == ord("N")) & valid
    qo = torch.zeros(n, qs, dtype=torch.uint8, device=dev)
    qo[:, :L] = torch.where(valid, qual | (isn.to(torch.uint8) << 7), torch.zeros_like(qual))
    lo = lens.to(torch.int16)  # 0..512 fits; same bits as uint16
    return so.contiguous(), qo.contiguous(), lo.contiguous()


def to_fastq_bytes(seq, qual, mate, name_prefix=b"@SIM:1:FC:1:1101"):
    """fixed-length reads only: vectorised FASTQ text (numpy), names zero padded"""
    import numpy as np
    seq = seq.cpu().numpy() if hasattr(seq, "cpu") else seq
    qual = qual.cpu().numpy() if hasattr(qual, "cpu") else qual
    n, L = seq.shape
    name0 = name_prefix + b":%09d %d:N:0:ATCG" % (0, mate)
    reclen = len(name0) + 1 + L + 3 + L + 1
    rec = np.zeros((n, reclen), dtype=np.uint8)
    rec[:, :len(name0)] = np.frombuffer(name0, dtype=np.uint8)
    digits = np.arange(n, dtype=np.int64)
    o = len(name_prefix) + 1
    for k in range(9):
        rec[:, o + 8 - k] = 48 + (digits // (10 ** k)) % 10
    p = len(name0)
    rec[:, p] = 10
    rec[:, p + 1:p + 1 + L] = seq
    rec[:, p + 1 + L] = 10
    rec[:, p + 2 + L] = ord("+")
    rec[:, p + 3 + L] = 10
    rec[:, p + 4 + L:p + 4 + 2 * L] = qual
    rec[:, p + 4 + 2 * L] = 10
    return rec.tobytes()


def to_fastq_tensor(seq, qual, mate, first=0, name_prefix=b"@SIM:1:FC:1:1101"):
    """to_fastq_bytes on the tensors' device: uint8 [n, record_bytes] (fixed-length reads, 9-digit names from `first`)"""
    n, L = seq.shape
    dev = seq.device
    name0 = name_prefix + b":%09d %d:N:0:ATCG" % (0, mate)
    p = len(name0)
    rec = torch.empty((n, p + 1 + L + 3 + L + 1), dtype=torch.uint8, device=dev)
    rec[:, :p] = torch.tensor(list(name0), dtype=torch.uint8, device=dev)
    idx = torch.arange(first, first + n, dtype=torch.int64, device=dev)
    o = len(name_prefix) + 1
    for k in range(9):
        rec[:, o + 8 - k] = (48 + (idx // (10 ** k)) % 10).to(torch.uint8)
    rec[:, p] = 10
    rec[:, p + 1:p + 1 + L] = seq
    rec[:, p + 1 + L] = 10
    rec[:, p + 2 + L] = ord("+")
    rec[:, p + 3 + L] = 10
    rec[:, p + 4 + L:p + 4 + 2 * L] = qual
    rec[:, p + 4 + 2 * L] = 10
    return rec
