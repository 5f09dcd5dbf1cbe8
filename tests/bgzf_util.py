"""BGZF writer for the tests (the format bgzip / htslib produce: RFC 1952 members with the BC extra subfield)"""
import struct
import zlib


def block(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY) -> bytes:
    assert len(data) <= 65536
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    payload = c.compress(data) + c.flush()
    bsize = 18 + len(payload) + 8
    assert bsize <= 65536 + 26
    hdr = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return hdr + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


EOF_BLOCK = block(b"")


def compress(text: bytes, block_bytes=0xff00, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, eof=True) -> bytes:
    out = [block(text[i:i + block_bytes], level, strategy) for i in range(0, len(text), block_bytes)]
    if eof:
        out.append(EOF_BLOCK)
    return b"".join(out)
