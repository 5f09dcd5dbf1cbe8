#!/usr/bin/env python3
"""Differential soak of the stream's host inflaters (fastp_amd/csrc/fq_pgunzip.h / fq_gunzip.h through fastp_gpu_stream_gunzip_file_mt, from the
emulator build of the library: host code) against zlib: random multi-member gzip streams (FASTQ text, random bytes, long runs; every level,
strategy, window and memLevel), a quarter of them with one bit flipped, each through a random geometry (1 - 8 threads, chunks of 700 bytes - 2 MiB,
random hand-over sizes).  A sound stream must give zlib's text; a damaged one must be an error unless zlib accepts it too (then the same text).

    python tools/gunzip_soak.py FIRST LAST
"""
import ctypes as C, os, sys, zlib, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import engines, synth
from fastp_amd import engine, abi
lib = engine.load_library(engines.build_sim())
lib.fastp_gpu_stream_gunzip_file_mt.argtypes = [C.c_char_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_int64)]
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "x.gz")
def member(data, level, strategy, wbits, memlevel):
    c = zlib.compressobj(level, zlib.DEFLATED, -wbits, memlevel, strategy)
    body = c.compress(data) + c.flush()
    return b"\x1f\x8b\x08\x00\0\0\0\0\x00\x03" + body + (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")
d = synth.synth_pairs(6000, L=150, seed=11, paired=False)
fq = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
t0 = time.time(); ok = bad = dmg_ok = 0
first, last = int(sys.argv[1]), int(sys.argv[2])
for seed in range(first, last):
    rng = np.random.default_rng(seed)
    parts = []
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 5))
        a = int(rng.integers(0, len(fq) - 1)); b = min(len(fq), a + int(rng.integers(1, 600000)))
        if kind <= 2: data = fq[a:b]
        elif kind == 3: data = rng.integers(0, 256, size=int(rng.integers(1, 90000)), dtype=np.uint8).tobytes()
        else: data = bytes([int(rng.integers(33, 80))]) * int(rng.integers(1, 200000)) + fq[a:a + 30000]
        parts.append((data, int(rng.integers(0, 10)), int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY])), int(rng.integers(9, 16)), int(rng.integers(1, 10))))
    blob = b"".join(member(*p) for p in parts)
    want = b"".join(p[0] for p in parts)
    damaged = rng.random() < 0.25 and len(blob) > 40
    if damaged:
        pos = int(rng.integers(12, len(blob) - 8))
        blob = blob[:pos] + bytes([blob[pos] ^ (1 << int(rng.integers(0, 8)))]) + blob[pos + 1:]
    open(path, "wb").write(blob)
    threads = int(rng.integers(1, 9)); chunk = int(rng.choice([700, 3000, 9000, 40000, 200000, 2 << 20]))
    out = np.zeros(len(want) + 100000, dtype=np.uint8); n = C.c_int64(0)
    rc = lib.fastp_gpu_stream_gunzip_file_mt(path.encode(), out.ctypes.data, out.size, int(rng.choice([0, 1000, 77777])), threads, chunk, C.byref(n))
    got = out[:n.value].tobytes()
    if not damaged:
        if rc == 0 and got == want: ok += 1
        else: bad += 1; print("FAIL seed", seed, rc, len(got), len(want), threads, chunk, flush=True)
    else:
        try:
            z = b""; dd = blob
            while dd:
                o = zlib.decompressobj(31); z += o.decompress(dd); 
                if not o.eof: raise zlib.error("cut")
                dd = o.unused_data
            zok = True
        except zlib.error:
            zok = False
        if zok:
            if rc == 0 and got == z: dmg_ok += 1
            else: bad += 1; print("FAIL (damaged, zlib accepts) seed", seed, rc, threads, chunk, flush=True)
        else:
            if rc != 0: dmg_ok += 1
            else: bad += 1; print("FAIL (damaged accepted) seed", seed, threads, chunk, flush=True)
print(f"seeds {first}..{last - 1}: {ok} sound streams equal to zlib's text, {dmg_ok} damaged streams handled as zlib handles them, {bad} FAILED, {time.time() - t0:.0f}s")
