"""fastp_amd/csrc/fq_gunzip.h and fq_pgunzip.h - the host inflaters the stream reads non-bgzip ".gz" inputs with (in place of
ISA-L's igzip behind FastqReader::readToBufIgzip, src/fastqreader.cpp:88-149; one thread / several threads on one stream) -
against zlib, through their C entries fastp_gpu_stream_gunzip_file / fastp_gpu_stream_gunzip_file_mt: every block type, compression level and strategy, several members, random and degenerate
data, arbitrary hand-over points; damaged streams must be errors.  Host code: runs from the emulator build of the library
in the CPU suite (no device needed either way)."""
import ctypes as C
import gzip
import os
import zlib

import numpy as np
import pytest

import engines
import synth
from fastp_amd import abi, engine


# (threads, chunk bytes): fq_gunzip.h alone; fq_pgunzip.h with chunks so small that every block of these files is a chunk of its
# own (they grow until a block fits), with chunks of a few blocks, and with the stream's own geometry
MODES = [(1, 0), (4, 1500), (4, 16000), (3, 40000), (1, 9000), (8, 2 << 20)]


class _Lib:
    def __init__(self, lib, threads, chunk):
        self.lib, self.threads, self.chunk = lib, threads, chunk


@pytest.fixture(scope="module", params=MODES, ids=lambda m: f"t{m[0]}_c{m[1]}")
def lib(request):
    lib = engine.load_library(engines.build_sim())
    lib.fastp_gpu_stream_gunzip_file.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    lib.fastp_gpu_stream_gunzip_file_mt.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_int64)]
    return _Lib(lib, *request.param)


def _gunzip(lib, tmp_path, blob: bytes, capacity: int, piece=0):
    p = os.path.join(str(tmp_path), "x.gz")
    with open(p, "wb") as f:
        f.write(blob)
    out = np.zeros(max(capacity, 1), dtype=np.uint8)
    n = C.c_int64(0)
    if lib.threads == 1 and lib.chunk == 0:
        rc = lib.lib.fastp_gpu_stream_gunzip_file(p.encode(), out.ctypes.data, capacity, piece, C.byref(n))
    else:
        rc = lib.lib.fastp_gpu_stream_gunzip_file_mt(p.encode(), out.ctypes.data, capacity, piece, lib.threads, lib.chunk, C.byref(n))
    return rc, out[:n.value].tobytes()


def _member(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15, memlevel=8, name=None, extra=None, comment=None, hcrc=False) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -wbits, memlevel, strategy)
    body = c.compress(data) + c.flush()
    flg = (4 if extra is not None else 0) | (8 if name is not None else 0) | (16 if comment is not None else 0) | (2 if hcrc else 0)
    hdr = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\x00\x03"
    if extra is not None:
        hdr += len(extra).to_bytes(2, "little") + extra
    if name is not None:
        hdr += name + b"\0"
    if comment is not None:
        hdr += comment + b"\0"
    if hcrc:
        hdr += (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little")
    return hdr + body + (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")


def _fastq(n, seed):
    d = synth.synth_pairs(n, L=150, seed=seed, paired=False)
    return synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)


def _datasets():
    rng = np.random.default_rng(12)
    fq = _fastq(4000, 5)
    return {
        "fastq": fq,
        "empty": b"",
        "one_byte": b"A",
        "zeros": bytes(300000),                                     # distance-1 matches of the maximum length
        "period3": b"ACG" * 70000,                                  # short distances, overlapping copies
        "period7": bytes(range(7)) * 40000,
        "random": rng.integers(0, 256, size=200000, dtype=np.uint8).tobytes(),      # stored blocks at any level
        "low_entropy": rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=400000, p=[0.3, 0.3, 0.19, 0.2, 0.01]).tobytes(),
        "long_codes": np.minimum(rng.geometric(0.02, size=300000), 255).astype(np.uint8).tobytes(),   # skewed: code lengths beyond the root table
        "text_then_random": fq[:150000] + rng.integers(0, 256, size=90000, dtype=np.uint8).tobytes() + fq[150000:260000],
    }


@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
def test_gunzip_equals_zlib_on_every_dataset_and_level(lib, tmp_path, level):
    for name, data in _datasets().items():
        rc, got = _gunzip(lib, tmp_path, _member(data, level), len(data) + 10)
        assert rc == 0 and got == data, (name, level, rc, len(got), len(data))


@pytest.mark.parametrize("strategy", [zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED])
def test_gunzip_strategies_windows_and_header_fields(lib, tmp_path, strategy):
    ds = _datasets()
    for name in ("fastq", "zeros", "low_entropy", "long_codes"):
        for wbits, memlevel in ((15, 8), (9, 1), (12, 9)):       # small windows / tiny blocks: many dynamic headers
            blob = _member(ds[name], 6, strategy, wbits, memlevel, name=b"reads.fq", extra=b"XY\x03\x00abc", comment=b"made by a test", hcrc=True)
            rc, got = _gunzip(lib, tmp_path, blob, len(ds[name]))
            assert rc == 0 and got == ds[name], (name, strategy, wbits, memlevel)


def test_gunzip_members_and_hand_over_points(lib, tmp_path):
    """several members (empty ones among them) and the text taken in pieces of every awkward size: the result does not depend
    on where the caller's buffers end (the stream hands over at trip boundaries), nor on the inflater's own 4 MiB / 1 MiB refills"""
    rng = np.random.default_rng(3)
    fq = _fastq(30000, 8)                                          # ~10 MB: several input and output refills
    cuts = [0] + sorted(int(x) for x in rng.integers(1, len(fq), size=9)) + [len(fq)]
    blob = b"".join(_member(fq[a:b], 1 + (i % 9)) for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])))
    blob = _member(b"") + blob[:len(blob) // 2] + blob[len(blob) // 2:] + _member(b"") + gzip.compress(b"tail\n")
    want = fq + b"tail\n"
    for piece in (0, 1, 7, 4096, 65537, 999983, len(want), len(want) + 5):
        if piece == 1 and len(want) > 2_000_000:
            rc, got = _gunzip(lib, tmp_path, blob, 3000, 1)        # (byte-wise on a prefix only: the buffer ends early -> E_OVERFLOW)
            assert rc == abi.E_OVERFLOW and got == want[:3000]
            continue
        rc, got = _gunzip(lib, tmp_path, blob, len(want), piece)
        assert rc == 0 and got == want, piece
    # members may not reach into each other: a distance that points in front of its member is an error even though the buffer
    # holds older text there
    raw = zlib.compressobj(9, zlib.DEFLATED, -15)
    body = raw.compress(b"0123456789" * 50) + raw.flush()
    second = zlib.compressobj(9, zlib.DEFLATED, -15, zdict=b"0123456789" * 50)
    body2 = second.compress(b"0123456789" * 50) + second.flush()     # its matches point into the dictionary = in front of the member
    good = _member(b"0123456789" * 50)
    data = b"0123456789" * 50
    bad = b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + body2 + (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + len(data).to_bytes(4, "little")
    rc, _ = _gunzip(lib, tmp_path, good + bad, 2000)
    assert rc == abi.E_INVALID


def test_gunzip_damaged_streams_are_errors(lib, tmp_path):
    rng = np.random.default_rng(21)
    fq = _fastq(3000, 9)
    blob = _member(fq, 6)
    n = len(fq)
    cases = {
        "cut inside the data": blob[:len(blob) // 2],
        "cut inside the trailer": blob[:-3],
        "crc": blob[:-8] + bytes([blob[-8] ^ 1]) + blob[-7:],
        "isize": blob[:-1] + bytes([blob[-1] ^ 0x40]),
        "not gzip": b"@read\nACGT\n+\nIIII\n" * 100,
        "wrong method": blob[:2] + b"\x07" + blob[3:],
        "reserved flag": blob[:3] + b"\x80" + blob[4:],
        "garbage behind": blob + b"this is no gzip header" * 3,
        "block type 3": b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + b"\x07" + b"\0" * 20,
        "stored LEN/NLEN": b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + b"\x01\x05\x00\xfa\xfe" + b"hello" + b"\0" * 8,
    }
    for what, data in cases.items():
        rc, _ = _gunzip(lib, tmp_path, data, n + 100)
        assert rc == abi.E_INVALID, what
    # bytes changed inside the deflate data: an error or (rarely) a CRC mismatch = an error too; never a crash, never success
    for k in range(60):
        pos = int(rng.integers(12, len(blob) - 8))
        data = blob[:pos] + bytes([blob[pos] ^ (1 << int(rng.integers(0, 8)))]) + blob[pos + 1:]
        rc, got = _gunzip(lib, tmp_path, data, n + 100)
        if rc == 0:     # a bit nobody reads (padding in front of a byte boundary): then zlib accepts the stream as well
            assert zlib.decompress(data, 31) == fq == got, (k, pos)
        else:           # (E_OVERFLOW: the damaged stream makes more text than the buffer takes before it breaks a rule)
            assert rc in (abi.E_INVALID, abi.E_OVERFLOW), (k, pos, rc)
    # over-subscribed and incomplete code sets in a dynamic header (hand-made bit streams)
    def bits(fields):
        acc, nb, out = 0, 0, bytearray()
        for v, w in fields:
            acc |= v << nb
            nb += w
            while nb >= 8:
                out.append(acc & 0xFF)
                acc >>= 8
                nb -= 8
        if nb:
            out.append(acc & 0xFF)
        return bytes(out)
    # BFINAL=1, BTYPE=2, HLIT=0 (257), HDIST=0 (1), HCLEN=15 (19): every code-length code 1 bit long -> over-subscribed
    over = bits([(1, 1), (2, 2), (0, 5), (0, 5), (15, 4)] + [(1, 3)] * 19)
    rc, _ = _gunzip(lib, tmp_path, b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + over + b"\0" * 16, 100)
    assert rc == abi.E_INVALID
    # ... only one code-length code of length 2 -> incomplete
    incomplete = bits([(1, 1), (2, 2), (0, 5), (0, 5), (15, 4)] + [(2, 3)] + [(0, 3)] * 18)
    rc, _ = _gunzip(lib, tmp_path, b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + incomplete + b"\0" * 16, 100)
    assert rc == abi.E_INVALID


def test_gunzip_random_streams_against_zlib(lib, tmp_path):
    """random mixtures of literals runs, repeats at every distance class and random bytes, random levels / strategies / windows"""
    rng = np.random.default_rng(77)
    for k in range(40):
        parts = []
        for _ in range(int(rng.integers(1, 30))):
            kind = int(rng.integers(0, 4))
            if kind == 0:
                parts.append(rng.integers(0, 256, size=int(rng.integers(1, 5000)), dtype=np.uint8).tobytes())
            elif kind == 1:
                parts.append(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 70000)))
            elif kind == 2 and parts:
                whole = b"".join(parts)
                d = int(rng.integers(1, min(len(whole), 32768) + 1))
                ln = int(rng.integers(3, 2000))
                seg = whole[-d:]
                parts.append((seg * (ln // len(seg) + 1))[:ln])
            else:
                parts.append(rng.choice(np.frombuffer(b"ACGT\n@+I", dtype=np.uint8), size=int(rng.integers(1, 30000))).tobytes())
        data = b"".join(parts)
        blob = _member(data, int(rng.integers(0, 10)), int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED])),
                       int(rng.integers(9, 16)), int(rng.integers(1, 10)))
        rc, got = _gunzip(lib, tmp_path, blob, len(data), int(rng.choice([0, 1000, 77777])))
        assert rc == 0 and got == data, k


def test_gunzip_block_headers_inside_stored_data_are_no_chunk_starts(lib, tmp_path):
    """a deflate stream stored inside another one (level 0: stored blocks): every block header of the inner stream parses as
    one where it lies, and is no block start of the outer stream - fq_pgunzip.h's chunks that begin there are thrown away when
    their predecessor does not end on their first bit; the text must not depend on it"""
    fq = _fastq(6000, 31)
    inner = _member(fq, 6, memlevel=2)                         # blocks of 256 symbols: thousands of headers
    data = b"@" + inner + fq[:50000] + inner[::-1]
    for outer_level, memlevel in ((0, 8), (1, 1), (6, 8)):
        rc, got = _gunzip(lib, tmp_path, _member(data, outer_level, memlevel=memlevel), len(data))
        assert rc == 0 and got == data, (outer_level, memlevel, rc, len(got))


def test_gunzip_match_in_front_of_the_member_found_in_a_later_chunk(lib, tmp_path):
    """the rule of test_gunzip_members_and_hand_over_points - no distance may reach in front of its member - where the offending
    match lies blocks behind the member's start: with several threads it is decoded with the window unknown and caught when the
    markers are resolved (the trailer would not tell: the bytes in front of the member are the dictionary's)"""
    rng = np.random.default_rng(5)
    dic = rng.integers(0, 256, size=30000, dtype=np.uint8).tobytes()
    first = _member(dic, 6)                                      # incompressible: stored blocks; its text = the dictionary
    for lead in (300, 4000, 20000):
        head = rng.integers(65, 70, size=lead, dtype=np.uint8).tobytes()
        tail = dic[20000:30000] + _fastq(300, 3)[:60000]            # matches into the dictionary (within 32 KiB), then text: a dynamic block
        data = head + tail
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zdict=dic)
        body = c.compress(head) + c.flush(zlib.Z_SYNC_FLUSH) + c.compress(tail) + c.flush()   # a block boundary in front of the matches
        bad = b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + body + (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + len(data).to_bytes(4, "little")
        with pytest.raises(zlib.error):
            zlib.decompressobj(-15).decompress(body)                # "invalid distance too far back"
        rc, _ = _gunzip(lib, tmp_path, first + bad, len(dic) + len(data) + 10)
        assert rc == abi.E_INVALID, lead
        # the same bytes as ONE member's continuation are fine: the control that the stream is otherwise sound
        d2 = zlib.decompressobj(-15, zdict=dic)
        assert d2.decompress(body) == data


def test_pgunzip_stream_geometry_on_a_file_of_many_chunks(tmp_path):
    """fq_pgunzip.h as the stream runs it (2 MiB chunks; and 1 MiB with another thread count) on ~50 MB of FASTQ text: a dozen
    chunks in several batches, two members, every hand-over between chunks, batches and members at production sizes"""
    lib = engine.load_library(engines.build_sim())
    lib.fastp_gpu_stream_gunzip_file_mt.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_int64)]
    d = synth.synth_pairs(150000, L=150, seed=41, paired=False)
    fq = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    cut = len(fq) * 2 // 3
    blob = _member(fq[:cut], 1) + _member(fq[cut:], 6)
    assert len(blob) > 5 * (2 << 20)
    for threads, chunk in ((8, 2 << 20), (3, 1 << 20), (1, 0)):   # (1, 0): fq_gunzip.h, whose 4 MiB input buffer this file refills a few times
        rc, got = _gunzip(_Lib(lib, threads, chunk), tmp_path, blob, len(fq), 16 << 20)
        assert rc == 0 and got == fq, (threads, chunk, rc, len(got))
    # a bit flipped far into the file: found whichever chunk it lands in
    pos = len(blob) * 3 // 5
    rc, _ = _gunzip(_Lib(lib, 8, 2 << 20), tmp_path, blob[:pos] + bytes([blob[pos] ^ 0x10]) + blob[pos + 1:], len(fq), 16 << 20)
    assert rc == abi.E_INVALID


def _sync_flush_members(text: bytes, pieces: int) -> bytes:
    """members whose deflate streams end in an empty stored block (Z_SYNC_FLUSH) followed by an empty fixed block (Z_FINISH
    with nothing pending): the few-bits-then-a-byte-wise-read endings of the one-thread inflater's refill path"""
    out, step = b"", max(1, len(text) // pieces)
    for a in range(0, len(text), step):
        part = text[a:a + step]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(part) + c.flush(zlib.Z_SYNC_FLUSH) + c.flush(zlib.Z_FINISH)
        out += b"\x1f\x8b\x08\0\0\0\0\0\x00\x03" + body + zlib.crc32(part).to_bytes(4, "little") + (len(part) & 0xffffffff).to_bytes(4, "little")
    return out


@pytest.mark.parametrize("incap_kb", [192, 256, 400, 0])
def test_gunzip_one_thread_refills_between_a_short_block_and_a_byte_wise_read(tmp_path, monkeypatch, incap_kb):
    """fq_gunzip.h with files LARGER than its input buffer (FASTP_GPU_STREAM_GUNZIP_INCAP_KB makes the buffer small; 0 = the
    production 4 MiB with a 12 MB file): after a refill has moved the unread input to the front, an end-of-block a few bits on
    and a stored block / trailer behind it (unread_bits stepping back over bytes the bit buffer still holds) must read the
    bytes that were in front of the refill point.  The round-4 advisor's reproduction (sync-flush + finish members)."""
    lib = engine.load_library(engines.build_sim())
    lib.fastp_gpu_stream_gunzip_file.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    if incap_kb:
        monkeypatch.setenv("FASTP_GPU_STREAM_GUNZIP_INCAP_KB", str(incap_kb))
    else:
        monkeypatch.delenv("FASTP_GPU_STREAM_GUNZIP_INCAP_KB", raising=False)
    d = synth.synth_pairs(150000 if not incap_kb else 12000, L=150, seed=43 + incap_kb, paired=False)
    fq = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    rng = np.random.default_rng(incap_kb)
    for pieces in (3, 40, 700):
        blob = _sync_flush_members(fq, pieces)
        assert len(blob) > (incap_kb << 10 if incap_kb else 2 * (4 << 20))
        assert gzip.decompress(blob) == fq
        rc, got = _gunzip(_Lib(lib, 1, 0), tmp_path, blob, len(fq), int(rng.integers(1 << 16, 1 << 22)))
        assert rc == 0 and got == fq, (incap_kb, pieces, rc, len(got))
    # stored members (level 0: every block is a byte-wise read) and a mixture, same geometry
    blob = b"".join(_member(fq[a:a + 70001], int(lv)) for a, lv in zip(range(0, len(fq), 70001), rng.integers(0, 3, size=len(fq) // 70001 + 1)))
    rc, got = _gunzip(_Lib(lib, 1, 0), tmp_path, blob, len(fq), 1 << 20)
    assert rc == 0 and got == fq


def test_gunzip_empty_file_is_no_gzip_stream(lib, tmp_path):
    """a 0-byte ".gz": FastqReader::init stops with "invalid gzip header" (fastqreader.cpp:193-196) - an error here as well"""
    rc, got = _gunzip(lib, tmp_path, b"", 16)
    assert rc == abi.E_INVALID and got == b""
