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

STREAM_GOLDENS = [n for n in golden_util.names() if "overlapped_out" not in n and n != "pe_exotic_default"]   # --overlapped_out's stream is the host glue's
SIM_CASES = ["pe_correction", "pe_merge_unmerged", "pe_filters", "pe_adapter_fasta", "pe_umi_per_read", "pe_overrep", "pe_noadapter_dedup",
             "se_adapter_cut", "se_adapter_fasta", "testdata_pe", "pe_exotic_merge", "pe_exotic_dedup_adapters", "se_exotic_adapter", "pe_exotic_overrep_merge"]


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
    want = [k for k in meta["outputs"] if k != "overlapped"]
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
