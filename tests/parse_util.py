"""helpers for the device FASTQ parser tests (hostsim: plain numpy buffers; GPU: torch tensors)"""
import numpy as np

from fastp_amd import abi, engine, hostloop


def expected(text: bytes, max_len: int, max_records=None, is_last=True):
    """what FastqReader + the packer would produce for the complete records of `text`:
    (seq rows, qual rows, lens, line offsets, line lengths, consumed)"""
    # split the reference's way: a line ends at the first \r or \n; \r\n is one terminator
    lines, offs = [], []
    i, n = 0, len(text)
    while i < n:
        j = i
        while j < n and text[j] not in (10, 13):
            j += 1
        if j == n:
            if not is_last:
                break
            lines.append(text[i:j]); offs.append(i); i = n
            break
        if text[j] == 13 and j == n - 1 and not is_last:
            break   # may be half of a \r\n
        lines.append(text[i:j]); offs.append(i)
        i = j + (2 if text[j] == 13 and j + 1 < n and text[j + 1] == 10 else 1)
        last_end = i
    nrec = len(lines) // 4
    if max_records is not None:
        nrec = min(nrec, max_records)
    ss, qs = abi.seq_stride(max_len), abi.qual_stride(max_len)
    seq = np.zeros((nrec, ss), dtype=np.uint8)
    qual = np.zeros((nrec, qs), dtype=np.uint8)
    lens = np.zeros(nrec, dtype=np.uint16)
    code = {65: 0, 84: 1, 67: 2, 71: 3}
    for r in range(nrec):
        s, q = lines[4 * r + 1], lines[4 * r + 3]
        lens[r] = len(s)
        for j, ch in enumerate(s):
            seq[r, j >> 2] |= code.get(ch, 0) << ((j & 3) * 2)
            qual[r, j] = q[j] | (0x80 if ch == 78 else 0)
    loff = np.array(offs[:4 * nrec], dtype=np.uint32)
    llen = np.array([len(x) for x in lines[:4 * nrec]], dtype=np.uint32)
    if nrec == 0:
        consumed = 0
    elif 4 * nrec < len(lines):
        consumed = offs[4 * nrec]
    else:
        consumed = i if 4 * nrec == len(lines) else offs[4 * nrec]
    return seq, qual, lens, loff, llen, consumed


def run_numpy(eng, text: bytes, max_len: int, max_records: int, is_last=True, check=True):
    """hostsim: 'device' pointers are host pointers"""
    pad = (-len(text)) % 16 + 16
    buf = np.frombuffer(text + b"\0" * pad, dtype=np.uint8).copy()
    base = buf.ctypes.data
    shift = (-base) % 16
    if shift:   # 16-byte alignment
        big = np.zeros(len(buf) + 16, dtype=np.uint8)
        o = (-big.ctypes.data) % 16
        big[o:o + len(buf)] = buf
        buf = big[o:o + len(buf)]
    ss, qs = abi.seq_stride(max_len), abi.qual_stride(max_len)
    seq = np.full((max(1, max_records), ss), 0xEE, dtype=np.uint8)
    qual = np.full((max(1, max_records), qs), 0xEE, dtype=np.uint8)
    lens = np.zeros(max(1, max_records), dtype=np.uint16)
    loff = np.zeros(4 * max(1, max_records), dtype=np.uint32)
    llen = np.zeros(4 * max(1, max_records), dtype=np.uint32)
    info = eng.parse_fastq(buf.ctypes.data, len(text), is_last, max_records, seq.ctypes.data, qual.ctypes.data,
                           lens.ctypes.data, loff.ctypes.data, llen.ctypes.data, check=check)
    n = info.n_records
    return info, seq[:n], qual[:n], lens[:n], loff[:4 * n], llen[:4 * n]
