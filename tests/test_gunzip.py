"""fastp_amd/csrc/fq_gunzip.h - the host inflater the stream reads non-bgzip ".gz" inputs with (in place of ISA-L's igzip
behind FastqReader::readToBufIgzip, src/fastqreader.cpp:88-149) - against zlib, through its C entry
fastp_gpu_stream_gunzip_file: every block type, compression level and strategy, several members, random and degenerate
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


@pytest.fixture(scope="module")
def lib():
    lib = engine.load_library(engines.build_sim())
    lib.fastp_gpu_stream_gunzip_file.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    return lib


def _gunzip(lib, tmp_path, blob: bytes, capacity: int, piece=0):
    p = os.path.join(str(tmp_path), "x.gz")
    with open(p, "wb") as f:
        f.write(blob)
    out = np.zeros(max(capacity, 1), dtype=np.uint8)
    n = C.c_int64(0)
    rc = lib.fastp_gpu_stream_gunzip_file(p.encode(), out.ctypes.data, capacity, piece, C.byref(n))
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
