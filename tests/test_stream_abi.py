"""include/fastp_gpu_stream.h through the C ABI (ctypes): FASTQ files in -> every output stream, the counter block and the
adapter maps out, against the committed golden fixtures (tests/golden/*.npz: inputs, md5 of every output file and the JSON
report of the REAL reference, made by tests/golden/make_golden.py).  No reference binary is needed: the CPU suite runs the
product sources on the SIMT emulator (small chunks: many trips, carried partial records), `-m gpu` the HIP library."""
import gzip
import os

import numpy as np
import pytest

import engines
import golden_util
import streamlib
from fastp_amd import abi, engine

STREAM_GOLDENS = [n for n in golden_util.names() if "overlapped_out" not in n and n != "pe_exotic_default"]
OVERLAPPED_GOLDENS = [n for n in golden_util.names() if "overlapped_out" in n]   # (their -m gpu runs: tests/test_zz_gpu_compressed_inputs.py)
SIM_CASES = golden_util.names()   # every golden: each -m gpu case of this file has its emulator twin (tests/conftest.py, the twin rule)


def _files(tmp_path, fq1, fq2):
    p1 = os.path.join(str(tmp_path), "in1.fq")
    open(p1, "wb").write(fq1)
    p2 = None
    if fq2 is not None:
        p2 = os.path.join(str(tmp_path), "in2.fq")
        open(p2, "wb").write(fq2)
    return p1, p2


def _synthetic(n, seed):
    import synth
    d = synth.synth_pairs(n, L=150, seed=seed)
    return synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1), synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2)


def _golden(lib, name, tmp_path, chunk_bytes, max_len=152, **kw):
    fq1, fq2, meta = golden_util.load(name)
    params = golden_util.params_for(name, max_len=max_len, fq1=fq1, fq2=fq2)
    p1, p2 = _files(tmp_path, fq1, fq2)
    want = list(meta["outputs"])     # ("overlapped": assembled on the host from the records, through the emit callback)
    if "out1" not in want:
        want += ["out1"] + (["out2"] if fq2 is not None else [])   # the reference opened them too (empty files)
    outs, ctr, lay, amaps, st = streamlib.run_files(lib, params, p1, p2, str(tmp_path), want=want, chunk_bytes=chunk_bytes,
                                                     umi=golden_util.umi_for(name), **kw)
    golden_util.check_against_golden(name, streamlib.as_outputs(outs, fq2 is not None), streamlib.report(ctr, lay, params, amaps), meta)
    return st


@pytest.mark.parametrize("name", SIM_CASES)
def test_sim_stream_equals_reference_golden(name, tmp_path):
    lib = engine.load_library(engines.build_sim())
    st = _golden(lib, name, tmp_path, chunk_bytes=60000)
    assert st.chunks >= 2 or name == "testdata_pe"


@pytest.mark.parametrize("name", ["pe_late_long_reads", "se_late_long_reads"])
def test_sim_stream_replans_like_the_reference_grows_its_buffers(name, tmp_path):
    """the goldens whose first 1100 units are at most 100 bases long and later ones 150: the REFERENCE sized its buffers from the
    first 1000 reads (Evaluator::computeSeqLen) and grew them (Stats::extendBuffer); the stream starts with that max_len,
    re-plans when the first longer read turns up, and must reproduce the reference's files and report (with -p: distance
    arrays of the evaluated length, sampling positions carried across the re-plan)"""
    lib = engine.load_library(engines.build_sim())
    st = _golden(lib, name, tmp_path, chunk_bytes=60000, max_len=100)
    assert st.replans >= 1 and st.max_len >= 150


def test_sim_stream_emit_callback_and_gz(tmp_path):
    """the emit callback (what feeds WriterThread::input) delivers the same bytes as the file descriptors; compressed streams
    are gzip members that inflate to the same text and end in bgzip's end-of-file member"""
    lib = engine.load_library(engines.build_sim())
    fq1, fq2, meta = golden_util.load("pe_default")
    params = golden_util.params_for("pe_default", max_len=152, fq1=fq1, fq2=fq2)
    p1, p2 = _files(tmp_path, fq1, fq2)
    a, ctr_a, lay, am_a, _ = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=50000)
    b, ctr_b, _, am_b, _ = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=90000, emit=True)
    c, ctr_c, _, _, st = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=70000, compress=("out1", "out2"))
    assert a == b and np.array_equal(ctr_a, ctr_b) and am_a.a1 == am_b.a1 and am_a.a2 == am_b.a2
    assert np.array_equal(ctr_a, ctr_c)
    for k in ("out1", "out2"):
        assert c[k][-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
        assert gzip.decompress(c[k]) == a[k]
    assert c["failed"] == a["failed"]
    assert st.bytes_out[0] == len(c["out1"])


def test_sim_stream_replans_and_limits(tmp_path):
    """max_len below the file's longest read: the stream re-plans and the result is that of a context that was large enough
    from the start; --reads_to_process cuts the stream; a malformed record ends it where FastqReader::read returns NULL"""
    lib = engine.load_library(engines.build_sim())
    fq1, fq2 = _synthetic(2200, seed=91)
    p1, p2 = _files(tmp_path, fq1, fq2)
    big = golden_util.params_for("pe_cut_right", max_len=152)
    small = golden_util.params_for("pe_cut_right", max_len=64)
    a, ctr_a, lay_a, am_a, st_a = streamlib.run_files(lib, big, p1, p2, str(tmp_path), chunk_bytes=60000)
    b, ctr_b, lay_b, am_b, st_b = streamlib.run_files(lib, small, p1, p2, str(tmp_path), chunk_bytes=60000)
    assert st_a.replans == 0 and st_b.replans >= 1 and st_b.max_len >= 150
    assert a == b and am_a.a1 == am_b.a1
    ra, rb = streamlib.report(ctr_a, lay_a, big, am_a), streamlib.report(ctr_b, lay_b, small, am_b)
    import refjson
    assert not refjson.diff(ra, rb)
    # --reads_to_process
    n = a["out1"].count(b"\n") // 4
    c, ctr_c, lay_c, _, st_c = streamlib.run_files(lib, big, p1, p2, str(tmp_path), chunk_bytes=60000, reads_to_process=777)
    assert st_c.units == 777 and ctr_c[lay_c.stats[0] + lay_c.st_reads] == 777
    assert a["out1"].startswith(c["out1"]) and 0 < len(c["out1"]) < len(a["out1"]) and n > 777
    # a record whose quality line is one character short, in the middle of file 2
    lines = fq2.split(b"\n")
    k = 4 * 1234 + 3
    lines[k] = lines[k][:-1]
    open(p2, "wb").write(b"\n".join(lines))
    d, ctr_d, lay_d, _, st_d = streamlib.run_files(lib, big, p1, p2, str(tmp_path), chunk_bytes=60000)
    assert st_d.truncated == 1 and st_d.units == 1234 and ctr_d[lay_d.stats[0] + lay_d.st_reads] == 1234
    # a quality character outside '!'..'~': refused with the record named, nothing silently skipped
    lines = fq1.split(b"\n")
    lines[4 * 50 + 3] = lines[4 * 50 + 3][:10] + b" " + lines[4 * 50 + 3][11:]
    open(p1, "wb").write(b"\n".join(lines))
    open(p2, "wb").write(fq2)
    with pytest.raises(streamlib.StreamError) as e:
        streamlib.run_files(lib, big, p1, p2, str(tmp_path), chunk_bytes=60000)
    assert e.value.code == abi.E_ALPHABET and "record 50 of file 1" in str(e.value)


def test_sim_stream_unequal_files_and_bad_arguments(tmp_path):
    lib = engine.load_library(engines.build_sim())
    fq1, fq2 = _synthetic(1500, seed=92)
    params = golden_util.params_for("pe_default", max_len=152)
    cut = b"\n".join(fq2.split(b"\n")[:4 * 901]) + b"\n"
    p1, p2 = _files(tmp_path, fq1, cut)
    outs, ctr, lay, _, st = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=60000)
    assert st.units == 901     # the reference processes min(read 1, read 2) records (peprocessor.cpp:363-370)
    with pytest.raises(streamlib.StreamError):
        streamlib.run_files(lib, params, p1, None, str(tmp_path))            # a paired engine needs two files
    with pytest.raises(streamlib.StreamError):
        streamlib.run_files(lib, params, os.path.join(str(tmp_path), "missing.fq"), p2, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.twin("test_sim_stream_equals_reference_golden")
@pytest.mark.parametrize("name", STREAM_GOLDENS)
def test_gpu_stream_equals_reference_golden(name, tmp_path):
    lib = engine.load_library()
    _golden(lib, name, tmp_path, chunk_bytes=1 << 20)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pe_late_long_reads", "se_late_long_reads"])
def test_gpu_stream_replans_like_the_reference_grows_its_buffers(name, tmp_path):
    lib = engine.load_library()
    st = _golden(lib, name, tmp_path, chunk_bytes=1 << 18, max_len=100)
    assert st.replans >= 1 and st.max_len >= 150


@pytest.mark.gpu
def test_gpu_stream_large_input_default_chunks(tmp_path):
    """300 000 pairs through the default 32 MiB chunks, plain and compressed, against the engine fed pack by pack"""
    import driver
    import synth
    lib = engine.load_library()
    n = 300000
    d = synth.synth_pairs(n, L=150, seed=77)
    fq1, fq2 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1), synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2)
    p1, p2 = _files(tmp_path, fq1, fq2)
    params = abi.default_params(True, 150)
    params.cut_right = 1
    outs, ctr, lay, amaps, st = streamlib.run_files(lib, params, p1, p2, str(tmp_path))
    eng = engines.gpu_engine(params)
    want, ctr2, rep2 = driver.run_engine(eng, params, fq1, fq2, pack=50000)
    eng.close()
    assert outs["out1"] == bytes(want.out1) and outs["out2"] == bytes(want.out2) and outs["failed"] == bytes(want.failed)
    assert np.array_equal(ctr, ctr2)
    import refjson
    assert not refjson.diff(rep2, streamlib.report(ctr, lay, params, amaps))
    gz, ctr3, _, _, _ = streamlib.run_files(lib, params, p1, p2, str(tmp_path), compress=("out1", "out2", "failed"))
    assert gzip.decompress(gz["out1"]) == outs["out1"] and gzip.decompress(gz["failed"]) == outs["failed"]
    assert np.array_equal(ctr, ctr3)


# ---- ".gz" inputs: bgzip-written files inflated on the device (in place of BgzfMtReader), other gzip streams by zlib on the host ----
def _pack(text: bytes, how: str, seed=0) -> bytes:
    import bgzf_util
    if how == "bgzf":
        return bgzf_util.compress(text, block_bytes=9000)
    if how == "bgzf_big":       # bgzip's own block size; no end-of-file member (htslib warns, readers go on)
        return bgzf_util.compress(text, block_bytes=0xff00, eof=False)
    if how == "gzip":
        return gzip.compress(text, 6)
    if how == "members":        # several gzip members cut at arbitrary bytes (what fastp's own WriterThread writes per pack)
        rng = np.random.default_rng(seed)
        cuts = sorted(set(int(x) for x in rng.integers(1, len(text), size=7))) + [len(text)]
        out, a = [], 0
        for b in cuts:
            out.append(gzip.compress(text[a:b], 1 + (a % 9)))
            a = b
        return b"".join(out)
    raise ValueError(how)


def _gz_files(tmp_path, fq1, fq2, how1, how2, tag=""):
    p1 = os.path.join(str(tmp_path), f"in1{tag}.fq.gz")
    open(p1, "wb").write(_pack(fq1, how1, 1))
    p2 = None
    if fq2 is not None:
        p2 = os.path.join(str(tmp_path), f"in2{tag}.fq.gz")
        open(p2, "wb").write(_pack(fq2, how2, 2))
    return p1, p2


def _golden_gz(lib, name, tmp_path, chunk_bytes, how1, how2, max_len=152):
    fq1, fq2, meta = golden_util.load(name)
    params = golden_util.params_for(name, max_len=max_len, fq1=fq1, fq2=fq2)
    p1, p2 = _gz_files(tmp_path, fq1, fq2, how1, how2)
    want = [k for k in meta["outputs"] if k != "overlapped"]
    if "out1" not in want:
        want += ["out1"] + (["out2"] if fq2 is not None else [])
    outs, ctr, lay, amaps, st = streamlib.run_files(lib, params, p1, p2, str(tmp_path), want=want, chunk_bytes=chunk_bytes, umi=golden_util.umi_for(name))
    golden_util.check_against_golden(name, streamlib.as_outputs(outs, fq2 is not None), streamlib.report(ctr, lay, params, amaps), meta)
    kinds = {"bgzf": 2, "bgzf_big": 2, "gzip": 1, "members": 1}
    assert st.input_kind[0] == kinds[how1] and (fq2 is None or st.input_kind[1] == kinds[how2])
    assert st.bytes_in[0] == len(fq1) or st.truncated or params.paired      # TEXT bytes; a paired run may stop inside the longer file
    assert 0 < st.bytes_file[0] <= os.path.getsize(p1)
    return st


GZ_CASES = [("pe_default", "bgzf", "bgzf"), ("pe_adapter_fasta", "bgzf", "gzip"), ("pe_merge_unmerged", "members", "bgzf"),
            ("se_adapter_cut", "bgzf", None), ("se_adapter_fasta", "members", None), ("pe_correction", "gzip", "members"),
            ("pe_exotic_merge", "bgzf", "bgzf"), ("pe_umi_per_read", "bgzf", "bgzf"),
            ("pe_overrep_merge", "bgzf_big", "bgzf_big"), ("pe_noadapter_dedup", "bgzf_big", "gzip")]
GZ_SIM_CASES = GZ_CASES   # every -m gpu case has its emulator twin (tests/conftest.py checks it at collection)


@pytest.mark.parametrize("name,how1,how2", GZ_SIM_CASES)
def test_sim_stream_compressed_inputs_equal_reference_golden(name, how1, how2, tmp_path):
    """the golden's input files compressed: what the stream writes and counts is what the reference made of the plain files
    (a decompressor changes nothing downstream), with several BGZF members and several trips per file"""
    lib = engine.load_library(engines.build_sim())
    # (bgzip-sized members hold more text than the 60 000-byte trips of the other cases)
    st = _golden_gz(lib, name, tmp_path, 100000 if "bgzf_big" in (how1, how2) else 60000, how1, how2)
    assert st.chunks >= 2


def test_sim_stream_bgzf_member_larger_than_a_trip_is_an_error_not_a_loop(tmp_path):
    """bgzip-sized members (65 280 bytes of text) and trips of 60 000: no trip can ever take the file's first member - an error
    that names the chunk size (round 5: with nothing carried yet the loop asked for the same empty trip for ever)"""
    lib = engine.load_library(engines.build_sim())
    fq1, fq2 = _synthetic(600, seed=97)
    params = golden_util.params_for("pe_cut_right", max_len=152)
    g1, g2 = _gz_files(tmp_path, fq1, fq2, "bgzf_big", "bgzf_big")
    with pytest.raises(streamlib.StreamError) as e:
        streamlib.run_files(lib, params, g1, g2, str(tmp_path), chunk_bytes=60000)
    assert "does not fit the chunk size" in str(e.value)


def test_sim_stream_bgzf_late_long_reads_replan(tmp_path):
    lib = engine.load_library(engines.build_sim())
    st = _golden_gz(lib, "pe_late_long_reads", tmp_path, 60000, "bgzf", "bgzf", max_len=100)
    assert st.replans >= 1


def _run_plain_and(lib, tmp_path, fq1, fq2, packed1, packed2, chunk_bytes=70000, **kw):
    params = golden_util.params_for("pe_cut_right", max_len=152)
    params.dup_enabled = 0
    p1, p2 = _files(tmp_path, fq1, fq2)
    a = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=chunk_bytes, **kw)
    g1, g2 = os.path.join(str(tmp_path), "x1.fq.gz"), os.path.join(str(tmp_path), "x2.fq.gz")
    open(g1, "wb").write(packed1)
    open(g2, "wb").write(packed2)
    b = streamlib.run_files(lib, params, g1, g2, str(tmp_path), chunk_bytes=chunk_bytes, **kw)
    return params, a, b


def _plain_gzip_geometries(lib, tmp_path, monkeypatch, n, chunk_bytes, geometries):
    """both inputs ordinary gzip streams of two members (levels 1 and 6), inflated by the stream's host threads (fq_pgunzip.h) in each
    of `geometries` = (threads, chunk KiB): every run == the run on the plain files, and the file was read to its end"""
    fq1, fq2 = _synthetic(n, seed=97)
    params = golden_util.params_for("pe_cut_right", max_len=152)
    p1, p2 = _files(tmp_path, fq1, fq2)
    a = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=chunk_bytes)
    two = lambda t: gzip.compress(t[:len(t) // 3], 1) + gzip.compress(t[len(t) // 3:], 6)   # noqa: E731
    g1, g2 = os.path.join(str(tmp_path), "pg1.fq.gz"), os.path.join(str(tmp_path), "pg2.fq.gz")
    open(g1, "wb").write(two(fq1))
    open(g2, "wb").write(two(fq2))
    for threads, kb in geometries:
        monkeypatch.setenv("FASTP_GPU_STREAM_GUNZIP_THREADS", str(threads))
        monkeypatch.setenv("FASTP_GPU_STREAM_GUNZIP_CHUNK_KB", str(kb))
        b = streamlib.run_files(lib, params, g1, g2, str(tmp_path), chunk_bytes=chunk_bytes)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[3].a1 == b[3].a1 and a[3].a2 == b[3].a2, (threads, kb)
        assert b[4].input_kind[0] == 1 and b[4].input_kind[1] == 1
        assert b[4].bytes_file[0] == os.path.getsize(g1) and b[4].bytes_in[0] == len(fq1) and b[4].bytes_in[1] == len(fq2), (threads, kb)


def test_sim_stream_plain_gzip_inputs_inflater_geometries(tmp_path, monkeypatch):
    lib = engine.load_library(engines.build_sim())
    _plain_gzip_geometries(lib, tmp_path, monkeypatch, 700, 60000, [(1, 2048), (2, 1), (5, 3), (8, 2048)])


def test_sim_stream_reads_its_own_compressed_output(tmp_path):
    """round trip: the ".gz" streams the device deflate writes are bgzip members - fed back as inputs they are inflated on the
    device and give the run its plain files give; bgzip-sized members (64 KiB of text) with the default trip size class"""
    import refjson
    lib = engine.load_library(engines.build_sim())
    fq1, fq2 = _synthetic(1200, seed=93)
    params = abi.default_params(True, 152)      # no trimming, no filtering that drops: out1/out2 are the inputs again
    params.adapter_trimming = 0
    params.quality_filter = 0
    params.length_filter = 0
    params.dup_enabled = 0
    p1, p2 = _files(tmp_path, fq1, fq2)
    first, _, _, _, _ = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=200000, compress=("out1", "out2"))
    if gzip.decompress(first["out1"]) != fq1:   # the parameter block filters after all: the round trip still holds on what came out
        fq1, fq2 = gzip.decompress(first["out1"]), gzip.decompress(first["out2"])
    params2, a, b = _run_plain_and(lib, tmp_path, fq1, fq2, first["out1"], first["out2"], chunk_bytes=200000)
    assert b[4].input_kind[0] == 2 and b[4].input_kind[1] == 2
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[3].a1 == b[3].a1 and a[3].a2 == b[3].a2
    assert not refjson.diff(streamlib.report(a[1], a[2], params2, a[3]), streamlib.report(b[1], b[2], params2, b[3]))


def test_sim_stream_compressed_inputs_limits_and_damage(tmp_path):
    """--reads_to_process and unequal files on compressed inputs; a file that ends inside a member, a member whose bytes were
    changed, and bytes that are no gzip header behind a member are errors (the reference: "igzip: unexpected eof" /
    "igzip: invalid gzip header found", fastqreader.cpp:102-146), never a shorter run"""
    import bgzf_util
    lib = engine.load_library(engines.build_sim())
    fq1, fq2 = _synthetic(800, seed=94)      # (the emulator runs the inflate kernel lane by lane: seconds per 100 KB)
    params = golden_util.params_for("pe_cut_right", max_len=152)
    params.dup_enabled = 0
    p1, p2 = _files(tmp_path, fq1, fq2)
    plain = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=60000, reads_to_process=333)
    for how in ("bgzf", "members"):
        g1, g2 = _gz_files(tmp_path, fq1, fq2, how, how, tag=how)
        got = streamlib.run_files(lib, params, g1, g2, str(tmp_path), chunk_bytes=60000, reads_to_process=333)
        assert got[0] == plain[0] and np.array_equal(got[1], plain[1]) and got[4].units == 333
    cut = b"\n".join(fq2.split(b"\n")[:4 * 401]) + b"\n"
    g1, g2 = _gz_files(tmp_path, fq1, cut, "bgzf", "bgzf", tag="u")
    assert streamlib.run_files(lib, params, g1, g2, str(tmp_path), chunk_bytes=60000)[4].units == 401
    good = {"bgzf": bgzf_util.compress(fq1, block_bytes=9000), "gzip": gzip.compress(fq1, 6)}
    g2 = p2                     # (the other file plain: a run may mix them)
    bad = os.path.join(str(tmp_path), "bad1.fq.gz")
    for kind, data in good.items():
        damaged = {"ends inside a member": data[:len(data) * 2 // 3],
                   "changed bytes": data[:len(data) // 2] + bytes(b ^ 0x5a for b in data[len(data) // 2:len(data) // 2 + 8]) + data[len(data) // 2 + 8:],
                   "no header behind a member": (data[:-28] if kind == "bgzf" else data) + b"this is not gzip" * 4}
        for what, blob in damaged.items():
            open(bad, "wb").write(blob)
            try:
                st = streamlib.run_files(lib, params, bad, g2, str(tmp_path), chunk_bytes=60000)[4]
            except streamlib.StreamError as e:
                assert e.code == abi.E_INVALID and ("gzip" in str(e) or "BGZF" in str(e)), (kind, what, str(e))
                continue
            # changed bytes inside a plain deflate stream may decode to something: then the text is no FASTQ and the stream
            # ends in front of the first malformed record, as FastqReader::read does (the member's CRC is never reached)
            assert kind == "gzip" and what == "changed bytes" and st.truncated == 1 and st.units < 800, (kind, what)
    # a BGZF member that does not fit a trip next to the carried text is named, not looped on
    open(bad, "wb").write(bgzf_util.compress(fq1, block_bytes=0xff00))
    with pytest.raises(streamlib.StreamError) as e:
        streamlib.run_files(lib, params, bad, g2, str(tmp_path), chunk_bytes=40000)
    assert "does not fit the chunk size" in str(e.value)


# ---- --interleaved_in: the mates' records alternate in one file ----
def _interleave(fq1: bytes, fq2: bytes) -> bytes:
    a, b = fq1.split(b"\n"), fq2.split(b"\n")
    out = []
    for i in range(0, min(len(a), len(b)) - 3, 4):
        out += a[i:i + 4] + b[i:i + 4]
    return b"\n".join(out) + b"\n"


def _golden_interleaved(lib, name, tmp_path, chunk_bytes, pack=None, max_len=152):
    fq1, fq2, meta = golden_util.load(name)
    params = golden_util.params_for(name, max_len=max_len, fq1=fq1, fq2=fq2)
    text = _interleave(fq1, fq2)
    p1 = os.path.join(str(tmp_path), "il.fq" + (".gz" if pack else ""))
    open(p1, "wb").write(_pack(text, pack, 5) if pack else text)
    want = [k for k in meta["outputs"] if k != "overlapped"]
    if "out1" not in want:
        want += ["out1", "out2"]
    outs, ctr, lay, amaps, st = streamlib.run_files(lib, params, p1, None, str(tmp_path), want=want, chunk_bytes=chunk_bytes, umi=golden_util.umi_for(name),
                                                     interleaved=True)
    golden_util.check_against_golden(name, streamlib.as_outputs(outs, True), streamlib.report(ctr, lay, params, amaps), meta)
    return st


IL_CASES = [("pe_default", None), ("pe_correction", None), ("pe_merge_unmerged", "bgzf"), ("pe_adapter_fasta", "gzip"), ("pe_exotic_dedup_adapters", None),
            ("pe_umi_per_read", None), ("pe_overrep", None), ("pe_filters", "members")]


@pytest.mark.parametrize("name,pack", IL_CASES)
def test_sim_stream_interleaved_input_equals_reference_golden(name, pack, tmp_path):
    """the paired goldens with their two input files dealt into ONE (read 1, read 2, read 1, ...): FastqReaderPair::read takes
    them in turn, so the run is the two-file run - outputs, counters, adapter maps; several trips, odd records carried"""
    lib = engine.load_library(engines.build_sim())
    st = _golden_interleaved(lib, name, tmp_path, 60000, pack)
    assert st.chunks >= 3


def test_sim_stream_interleaved_limits_odd_tail_and_replan(tmp_path):
    lib = engine.load_library(engines.build_sim())
    fq1, fq2 = _synthetic(1300, seed=95)
    params = golden_util.params_for("pe_cut_right", max_len=152)
    params.dup_enabled = 0
    p1, p2 = _files(tmp_path, fq1, fq2)
    two = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=60000)
    il = os.path.join(str(tmp_path), "il.fq")
    text = _interleave(fq1, fq2)
    open(il, "wb").write(text)
    one = streamlib.run_files(lib, params, il, None, str(tmp_path), chunk_bytes=60000, interleaved=True)
    assert one[0] == two[0] and np.array_equal(one[1], two[1]) and one[3].a1 == two[3].a1 and one[3].a2 == two[3].a2 and one[4].units == 1300
    # a record without its mate at the end: the reference's last pair is incomplete = its end of input (peprocessor.cpp:906-909)
    open(il, "wb").write(text + b"\n".join(fq1.split(b"\n")[:4]) + b"\n")
    odd = streamlib.run_files(lib, params, il, None, str(tmp_path), chunk_bytes=60000, interleaved=True)
    assert odd[0] == two[0] and np.array_equal(odd[1], two[1])
    # --reads_to_process counts pairs
    lim2 = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=60000, reads_to_process=555)
    open(il, "wb").write(text)
    lim1 = streamlib.run_files(lib, params, il, None, str(tmp_path), chunk_bytes=60000, reads_to_process=555, interleaved=True)
    assert lim1[0] == lim2[0] and np.array_equal(lim1[1], lim2[1]) and lim1[4].units == 555
    # a malformed record (either mate's) ends the stream in front of its pair
    lines = text.split(b"\n")
    k = 4 * (2 * 700 + 1) + 3           # the quality line of read 2 of pair 700
    lines[k] = lines[k][:-1]
    open(il, "wb").write(b"\n".join(lines))
    cut = streamlib.run_files(lib, params, il, None, str(tmp_path), chunk_bytes=60000, interleaved=True)
    assert cut[4].truncated == 1 and cut[4].units == 700
    # a longer read later in the file: re-plan with the dealt-out rows reallocated
    small = golden_util.params_for("pe_cut_right", max_len=64)
    small.dup_enabled = 0
    open(il, "wb").write(text)
    rp = streamlib.run_files(lib, small, il, None, str(tmp_path), chunk_bytes=60000, interleaved=True)
    assert rp[4].replans >= 1 and rp[0] == two[0]
    with pytest.raises(streamlib.StreamError):
        streamlib.run_files(lib, params, il, p2, str(tmp_path), interleaved=True)     # interleaved input is ONE file


def test_sim_stream_phred64_input(tmp_path):
    """--phred64 through the C ABI: the run on phred+64 text with config.phred64 equals the run on the same reads written as
    phred+33 (convertPhred64To33: max(33, q - 31), so qualities below '@' come out as '!'), two files and interleaved"""
    import synth
    lib = engine.load_library(engines.build_sim())
    d = synth.synth_pairs(900, L=150, seed=96, insert_mean=120.0, insert_sd=30.0)   # inserts shorter than the reads: read-through
    rng = np.random.default_rng(4)
    q64 = {}
    for m in ("1", "2"):
        q = d["qual" + m]
        body = q >= 33
        hi = q.copy()
        hi[body] = np.minimum(q[body] + 31, 126)
        low = body & (rng.random(q.shape) < 0.01)
        hi[low] = rng.integers(59, 64, size=int(low.sum())).astype(q.dtype)
        q64[m] = hi
        conv = hi.copy()
        conv[body] = np.maximum(33, hi[body].astype(np.int32) - 31).astype(q.dtype)
        d["qual" + m] = conv                       # what the reference's reader hands on
    a1, a2 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1), synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2)
    b1, b2 = synth.to_fastq(d["seq1"], q64["1"], d["len1"], 1), synth.to_fastq(d["seq2"], q64["2"], d["len2"], 2)
    params = golden_util.params_for("pe_cut_right", max_len=152)
    params.dup_enabled = 0
    p1, p2 = _files(tmp_path, a1, a2)
    want = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=60000)
    p1, p2 = _files(tmp_path, b1, b2)
    got = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=60000, phred64=True)
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and got[3].a1 == want[3].a1
    il = os.path.join(str(tmp_path), "il64.fq")
    open(il, "wb").write(_interleave(b1, b2))
    got = streamlib.run_files(lib, params, il, None, str(tmp_path), chunk_bytes=60000, phred64=True, interleaved=True)
    assert got[0] == want[0] and np.array_equal(got[1], want[1])
    # --overlapped_out's stream is assembled on the host, from the file's text: its qualities are converted there as well
    # (binding fuzz seed 3821)
    ov = golden_util.params_for("pe_overlapped_out_noadapter", max_len=152)   # (no adapter trimming: read 1 reaches past the overlap)
    ov.dup_enabled = 0
    ov.correction = 1
    streams = ("out1", "out2", "failed", "overlapped")
    p1, p2 = _files(tmp_path, a1, a2)
    want = streamlib.run_files(lib, ov, p1, p2, str(tmp_path), chunk_bytes=60000, want=streams)
    p1, p2 = _files(tmp_path, b1, b2)
    got = streamlib.run_files(lib, ov, p1, p2, str(tmp_path), chunk_bytes=60000, want=streams, phred64=True)
    body = [ln for ln in want[0]["overlapped"].split(b"\n")[3::4] if ln]
    assert len(body) > 100 and got[0] == want[0] and np.array_equal(got[1], want[1])


def test_sim_stream_reads_pipes(tmp_path):
    """inputs that are not regular files (--stdin, FIFOs): read in sequence, plain and gzip (a bgzip-written stream through a pipe
    takes the host inflater: a pipe cannot be looked into twice); same result as the files"""
    import threading
    lib = engine.load_library(engines.build_sim())
    fq1, fq2 = _synthetic(900, seed=97)
    params = golden_util.params_for("pe_cut_right", max_len=152)
    params.dup_enabled = 0
    p1, p2 = _files(tmp_path, fq1, fq2)
    want = streamlib.run_files(lib, params, p1, p2, str(tmp_path), chunk_bytes=60000)

    def through_pipes(blob1, blob2, suffix):
        f1, f2 = os.path.join(str(tmp_path), "pipe1.fq" + suffix), os.path.join(str(tmp_path), "pipe2.fq" + suffix)
        for f in (f1, f2):
            if os.path.exists(f):
                os.remove(f)
            os.mkfifo(f)

        def feed(path, blob):
            with open(path, "wb") as w:
                for a in range(0, len(blob), 7001):      # in dribbles: short reads on the other side
                    w.write(blob[a:a + 7001])
        ts = [threading.Thread(target=feed, args=(f1, blob1)), threading.Thread(target=feed, args=(f2, blob2))]
        for t in ts:
            t.start()
        try:
            return streamlib.run_files(lib, params, f1, f2, str(tmp_path), chunk_bytes=60000)
        finally:
            for t in ts:
                t.join()
    got = through_pipes(fq1, fq2, "")
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and got[4].units == 900
    import bgzf_util
    got = through_pipes(gzip.compress(fq1, 4), bgzf_util.compress(fq2, block_bytes=9000), ".gz")
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and list(got[4].input_kind) == [1, 1]
