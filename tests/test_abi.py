"""CPU-side checks of the C-ABI shared library: it loads, exports every symbol of
include/fastp_gpu.h, and its device-free entry points behave (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oraclelib
from fastp_amd import abi, engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    return engine.load_library()


def test_library_exports_every_declared_symbol(lib):
    # every header under include/: the engine (fastp_gpu.h), the host glue (fastp_gpu_host.h), the file stream (fastp_gpu_stream.h)
    declared = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)   # (comments mention calls such as WriterThread::input(...))
        declared |= set(re.findall(r"\b(fastp_gpu_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"fastp_gpu_stream_emit_fn"}   # a function-pointer type, not an entry point
    assert declared, "no declarations parsed"
    assert declared == set(engine.EXPORTS), declared ^ set(engine.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"


def test_struct_sizes_match_header():
    assert C.sizeof(abi.ReadResult) == 12
    assert C.sizeof(abi.PairResult) == 8
    assert C.sizeof(abi.Correction) == 8
    assert C.sizeof(abi.AdapterEvent) == 12


def test_default_params_match_python_mirror(lib):
    for paired in (0, 1):
        p = abi.Params()
        lib.fastp_gpu_default_params(C.byref(p), paired, 150)
        q = abi.default_params(paired, 150)
        for name, _ in abi.Params._fields_:
            if name in ("reserved", "adapter_seq_r1", "adapter_seq_r2", "adapter_fasta", "overrep_seqs1", "overrep_seqs2"):
                continue
            assert getattr(p, name) == getattr(q, name), name


def test_counter_layout_matches_oracle(lib):
    for cycles, isz in ((150, 512), (300, 512), (251, 100)):
        a = abi.CounterLayout()
        lib.fastp_gpu_counter_layout_for(cycles, isz, C.byref(a))
        b = oraclelib.layout(cycles, isz)
        assert bytes(a) == bytes(b)


def test_strides(lib):
    for ml in (1, 4, 31, 32, 33, 100, 150, 151, 250, 512):
        assert lib.fastp_gpu_seq_stride(ml) == abi.seq_stride(ml)
        assert lib.fastp_gpu_qual_stride(ml) == abi.qual_stride(ml)
        assert lib.fastp_gpu_seq_stride(ml) * 4 >= ml and lib.fastp_gpu_seq_stride(ml) % 8 == 0


def _np_pack(seq, qual, lens, max_len):
    """independent numpy restatement of the packed layout documented in fastp_gpu.h"""
    code = np.zeros(256, dtype=np.uint8)
    for ch, c in zip(b"ATCG", range(4)):
        code[ch] = c
    n = len(lens)
    ss, qs = abi.seq_stride(max_len), abi.qual_stride(max_len)
    so = np.zeros((n, ss), dtype=np.uint8)
    qo = np.zeros((n, qs), dtype=np.uint8)
    for i in range(n):
        L = int(lens[i])
        s = seq[i, :L]
        c = code[s].astype(np.uint32)
        for j in range(L):
            so[i, j >> 2] |= c[j] << ((j & 3) * 2)
        qo[i, :L] = qual[i, :L] | np.where(s == ord("N"), 0x80, 0).astype(np.uint8)
    return so, qo, lens.astype(np.uint16)


def test_pack_reads_layout_and_errors(lib):
    import synth
    d = synth.synth_pairs(300, L=150, seed=3)
    so, qo, lo = engine.pack_ascii(lib, 150, d["seq1"], d["qual1"], d["len1"])
    eso, eqo, elo = _np_pack(d["seq1"], d["qual1"], d["len1"], 150)
    assert np.array_equal(so, eso) and np.array_equal(qo, eqo) and np.array_equal(lo, elo)
    bad = d["seq1"].copy()
    bad[7, 3] = ord("R")
    with pytest.raises(engine.EngineError) as e:
        engine.pack_ascii(lib, 150, bad, d["qual1"], d["len1"])
    assert e.value.code == abi.E_ALPHABET
    # quality characters outside '!'..'~' are refused too (the kernels take q - 33 as an unsigned counter field)
    for qbad in (32, 10, 127, 200):
        badq = d["qual1"].copy()
        badq[11, 5] = qbad
        with pytest.raises(engine.EngineError) as e:
            engine.pack_ascii(lib, 150, d["seq1"], badq, d["len1"])
        assert e.value.code == abi.E_ALPHABET, qbad
    with pytest.raises(engine.EngineError) as e:
        engine.pack_ascii(lib, 100, d["seq1"], d["qual1"], d["len1"])
    assert e.value.code == abi.E_TOO_LONG


def _pack_rows(lib, seqs, quals, max_len):
    n = len(seqs)
    lib.fastp_gpu_seq_stride.restype = C.c_size_t
    lib.fastp_gpu_qual_stride.restype = C.c_size_t
    ss, qs = lib.fastp_gpu_seq_stride(max_len), lib.fastp_gpu_qual_stride(max_len)
    so = np.full(n * ss, 0xEE, dtype=np.uint8)     # (stale bytes: the packer must write every byte of a row)
    qo = np.full(n * qs, 0xEE, dtype=np.uint8)
    lo = np.zeros(n, dtype=np.uint16)
    sp, qp = (C.c_char_p * n)(*seqs), (C.c_char_p * n)(*quals)
    lens, bad = (C.c_int32 * n)(*[len(s) for s in seqs]), C.c_int32(-1)
    rc = lib.fastp_gpu_pack_reads(max_len, n, sp, qp, lens, so.ctypes.data_as(C.c_void_p), qo.ctypes.data_as(C.c_void_p),
                                  lo.ctypes.data_as(C.c_void_p), C.byref(bad))
    return rc, bad.value, so.reshape(n, ss), qo.reshape(n, qs), lo


def test_pack_reads_eight_at_a_time_equals_byte_by_byte(lib):
    """the packer works on groups of eight characters: every read length around the group and stride boundaries against a
    byte-by-byte restatement, and every foreign letter / quality value at every position of a group refused"""
    rng = np.random.default_rng(3)
    code = {ord("A"): 0, ord("T"): 1, ord("C"): 2, ord("G"): 3, ord("N"): 0}
    for max_len in (1, 7, 8, 9, 31, 32, 33, 150, 151, 250, 256):
        seqs, quals = [], []
        for _ in range(120):
            L = int(rng.integers(0, max_len + 1))
            seqs.append(bytes(rng.choice(list(b"ACGTN"), size=L, p=[.24, .24, .24, .24, .04]).astype(np.uint8)))
            quals.append(bytes(rng.integers(33, 127, size=L).astype(np.uint8)))
        rc, bad, so, qo, lo = _pack_rows(lib, seqs, quals, max_len)
        assert rc == 0, (max_len, rc, bad)
        es, eq = np.zeros_like(so), np.zeros_like(qo)
        for i, (s, q) in enumerate(zip(seqs, quals)):
            for j, (c, qc) in enumerate(zip(s, q)):
                es[i, j >> 2] |= code[c] << (2 * (j & 3))
                eq[i, j] = qc | (0x80 if c == ord("N") else 0)
        assert np.array_equal(so, es) and np.array_equal(qo, eq) and list(lo) == [len(s) for s in seqs], max_len
    good = b"ACGTNACGTNACGTNACGTNACG"
    for pos in (0, 1, 6, 7, 8, 9, 15, 16, 22):
        for c in range(1, 256):
            if c in b"ACGTN":
                continue
            s = bytearray(good)
            s[pos] = c
            rc, bad, *_ = _pack_rows(lib, [b"ACGT", bytes(s)], [b"IIII", b"I" * len(s)], 40)
            assert rc == abi.E_ALPHABET and bad == 1, (pos, c, rc, bad)
        for qc in list(range(1, 33)) + list(range(127, 256)):
            q = bytearray(b"I" * len(good))
            q[pos] = qc
            rc, bad, *_ = _pack_rows(lib, [good], [bytes(q)], 40)
            assert rc == abi.E_ALPHABET and bad == 0, (pos, qc, rc)


def test_host_glue_library_exports_every_declared_symbol(lib):
    """include/fastp_gpu_host.h (the C++ string side of the patched worker loop)"""
    import cpphost
    hdr = open(os.path.join(ROOT, "include", "fastp_gpu_host.h")).read()
    declared = set(re.findall(r"\b(fastp_gpu_host_[a-z_]+)\s*\(", hdr))
    assert declared == set(cpphost.EXPORTS), declared ^ set(cpphost.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"


def test_headers_compile_as_plain_c_and_cpp(tmp_path):
    """include/fastp_gpu.h is a C header (the boundary is a C ABI); fastp_gpu_host.h is the C++ glue header"""
    import subprocess
    inc = os.path.join(ROOT, "include")
    c = tmp_path / "t.c"
    c.write_text('#include "fastp_gpu.h"\nint main(void){fastp_gpu_params p; fastp_gpu_default_params(&p,1,150); return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(c)])
    cpp = tmp_path / "t.cpp"
    cpp.write_text('#include "fastp_gpu_host.h"\nint main(){return 0;}\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", inc, "-fsyntax-only", str(cpp)])
