"""The `-m gpu` side of the compressed-input and interleaved-input tests (".gz" inputs of the stream: bgzip-written files inflated on the device in
place of BgzfMtReader, other gzip streams by zlib inside the stream).  Their CPU-suite counterparts run on the emulator in
tests/test_stream_abi.py and tests/test_ref_binding.py; these were written in a session without GPU minutes and live in a file
of their own, collected last, so that nothing else's result depends on them."""
import gzip
import os

import numpy as np
import pytest

import streamlib
import test_ref_binding as rb
from fastp_amd import abi, engine
from test_stream_abi import GZ_CASES, IL_CASES, OVERLAPPED_GOLDENS, _files, _golden, _golden_gz, _golden_interleaved, _plain_gzip_geometries, _run_plain_and


@pytest.mark.gpu
@pytest.mark.twin("test_sim_stream_compressed_inputs_equal_reference_golden")
@pytest.mark.parametrize("name,how1,how2", GZ_CASES)
def test_gpu_stream_compressed_inputs_equal_reference_golden(name, how1, how2, tmp_path):
    lib = engine.load_library()
    _golden_gz(lib, name, tmp_path, 1 << 20, how1, how2)


@pytest.mark.gpu
def test_gpu_stream_large_bgzf_input_default_chunks(tmp_path):
    """300 000 pairs as bgzip-sized members through the default chunks (thousands of members per launch: both inflate kernels'
    ranges), as plain gzip, and the stream's own compressed output fed back: all equal the run on the plain files"""
    import bgzf_util
    import synth
    lib = engine.load_library()
    d = synth.synth_pairs(300000, L=150, seed=78)
    fq1, fq2 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1), synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2)
    params = abi.default_params(True, 150)
    params.cut_right = 1
    p1, p2 = _files(tmp_path, fq1, fq2)
    a = streamlib.run_files(lib, params, p1, p2, str(tmp_path))
    z = streamlib.run_files(lib, params, p1, p2, str(tmp_path), compress=("out1", "out2"))
    for pack1, pack2 in ((bgzf_util.compress(fq1, level=1), bgzf_util.compress(fq2, level=1)), (gzip.compress(fq1, 1), bgzf_util.compress(fq2, level=1))):
        g1, g2 = os.path.join(str(tmp_path), "big1.fq.gz"), os.path.join(str(tmp_path), "big2.fq.gz")
        open(g1, "wb").write(pack1)
        open(g2, "wb").write(pack2)
        b = streamlib.run_files(lib, params, g1, g2, str(tmp_path))
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[3].a1 == b[3].a1
        assert 0.9 * len(pack1) <= b[4].bytes_file[0] <= len(pack1) and b[4].bytes_in[0] == len(fq1)
    # the stream's own ".gz" outputs as inputs of a second run == that run on their text
    t1, t2 = gzip.decompress(z[0]["out1"]), gzip.decompress(z[0]["out2"])
    _, c, e = _run_plain_and(lib, tmp_path, t1, t2, z[0]["out1"], z[0]["out2"], chunk_bytes=0)
    assert c[0] == e[0] and np.array_equal(c[1], e[1]) and e[4].input_kind[0] == 2


@pytest.mark.gpu
def test_gpu_stream_plain_gzip_inputs_inflater_geometries(tmp_path, monkeypatch):
    """120 000 pairs as two ordinary gzip files: the stream's several-threads host inflater in its production geometry (2 MiB
    chunks: half a dozen per file), with more threads than chunks, with one thread (fq_gunzip.h) and with small chunks"""
    lib = engine.load_library()
    _plain_gzip_geometries(lib, tmp_path, monkeypatch, 120000, 0, [(8, 2048), (16, 2048), (1, 2048), (3, 64)])


@pytest.mark.gpu
@pytest.mark.twin("test_patched_reference_compressed_inputs")
@pytest.mark.parametrize("name,how,kw", rb.GZ_BINDING_CASES)
def test_gpu_patched_reference_compressed_inputs(name, how, kw, tmp_path):
    if not (os.path.exists(rb.REF) and os.path.exists(rb.REF_GPU)):
        pytest.skip("oracle/_ref binaries did not travel to this box")
    err = rb._check(name, rb.REF_GPU, 30000, tmp_path, seed=44, gz_in=how, **kw)
    assert ("inflated on the device" in err) == ("bgzf" in how), err[-800:]


@pytest.mark.gpu
@pytest.mark.twin("test_sim_stream_interleaved_input_equals_reference_golden")
@pytest.mark.parametrize("name,pack", IL_CASES)
def test_gpu_stream_interleaved_input_equals_reference_golden(name, pack, tmp_path):
    lib = engine.load_library()
    _golden_interleaved(lib, name, tmp_path, 1 << 20, pack)


@pytest.mark.gpu
@pytest.mark.twin("test_patched_reference_interleaved_input")
@pytest.mark.parametrize("name,kw", rb.IL_BINDING_CASES)
def test_gpu_patched_reference_interleaved_input(name, kw, tmp_path):
    if not (os.path.exists(rb.REF) and os.path.exists(rb.REF_GPU)):
        pytest.skip("oracle/_ref binaries did not travel to this box")
    rb._check(name, rb.REF_GPU, 30000, tmp_path, seed=45, interleaved=True, **kw)


@pytest.mark.gpu
@pytest.mark.twin("test_patched_reference_phred64_input")
@pytest.mark.parametrize("name,kw", rb.PHRED64_CASES)
def test_gpu_patched_reference_phred64_input(name, kw, tmp_path):
    if not (os.path.exists(rb.REF) and os.path.exists(rb.REF_GPU)):
        pytest.skip("oracle/_ref binaries did not travel to this box")
    # 30 000 units against the reference run on the converted qualities (its own --phred64 leaves the reads that come back
    # from its ReadPool unconverted, rb._check), one pack of 900 against `fastp_ref --phred64` itself
    rb._check(name, rb.REF_GPU, 30000, tmp_path, seed=46, mutate=rb._to_phred64, more_flags=("--phred64",), ref_on_phred33=True, **kw)
    sub = tmp_path / "one_pack"
    sub.mkdir()
    rb._check(name, rb.REF_GPU, 900, sub, seed=49, mutate=rb._to_phred64, more_flags=("--phred64",), **kw)


@pytest.mark.gpu
@pytest.mark.twin("test_patched_reference_stdin_input")
@pytest.mark.parametrize("name,kw", rb.STDIN_CASES)
def test_gpu_patched_reference_stdin_input(name, kw, tmp_path):
    if not (os.path.exists(rb.REF) and os.path.exists(rb.REF_GPU)):
        pytest.skip("oracle/_ref binaries did not travel to this box")
    rb._check(name, rb.REF_GPU, 30000, tmp_path, seed=47, stdin_pipe=True, **kw)


@pytest.mark.gpu
@pytest.mark.twin("test_sim_stream_equals_reference_golden")
@pytest.mark.parametrize("name", OVERLAPPED_GOLDENS)
def test_gpu_stream_overlapped_out_equals_reference_golden(name, tmp_path):
    """--overlapped_out in stream mode: six streams from the device formatter, the seventh assembled on the host of the loop"""
    lib = engine.load_library()
    _golden(lib, name, tmp_path, chunk_bytes=1 << 20)


@pytest.mark.gpu
@pytest.mark.twin("test_patched_reference_on_emulator_equals_reference")
@pytest.mark.parametrize("name", [n for n in rb.BINDING_CASES if rb._overlapped_out(n)])
def test_gpu_patched_reference_overlapped_out_stream_mode(name, tmp_path):
    if not (os.path.exists(rb.REF) and os.path.exists(rb.REF_GPU)):
        pytest.skip("oracle/_ref binaries did not travel to this box")
    rb._check(name, rb.REF_GPU, 30000, tmp_path, seed=48, threads=3)
