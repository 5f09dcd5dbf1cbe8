"""The drop-in boundary end to end: REAL reference fastp (its CLI, reader / writer threads, Stats / FilterResult
objects, JSON reporter) with the two worker-loop bodies bound to the engine by oracle/patches/gpu_worker.cpp
(oracle/build_ref_gpu.sh -> oracle/_ref/fastp_ref_gpu), against the unpatched reference `fastp_ref -w 1` on the same
files: every output FASTQ byte for byte, and the JSON report fastp ITSELF wrote, value for value.

The CPU suite runs the binding linked against the SIMT-emulator build of the engine (fastp_ref_gpusim, small
inputs); the `-m gpu` test runs the real library.  Both binaries are built in the build container (they need
/root/reference) and travel to the GPU box with oracle/_ref/."""
import json
import os
import subprocess

import pytest

import cases
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")
REF_GPU = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu")
REF_SIM = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpusim")

# case -> extra CLI flags that name the secondary outputs the case exercises
BINDING_CASES = {
    "pe_default": [],
    "pe_cut_right": [],
    "pe_cut_front_tail": [],
    "pe_polyg_polyx": [],
    "pe_adapter_seq": [],
    "pe_correction": [],
    "pe_trim_fixed": [],
    "pe_filters": ["--unpaired1", "@TMP@/u1.fq", "--unpaired2", "@TMP@/u2.fq"],
    "pe_noadapter_dedup": [],
    "pe_nofilters": [],
    "pe_merge": [],
    "pe_merge_unmerged": [],
    "pe_allow_gap_indel": [],
    "pe_umi_per_read": [],
    "pe_adapter_fasta": [],
    "pe_overrep": [],
    "pe_overlapped_out_trims": [],
    "pe_overlapped_out_noadapter": [],
    "se_default_noadapter": [],
    "se_adapter_cut": [],
    "se_umi_read1": [],
    "se_adapter_indel": [],
    "se_overrep": [],
    "pe_adapter_long": [],
    "se_adapter_long_indel": [],
}


_BUILT = False


def _ensure_built():
    global _BUILT
    if not _BUILT and os.path.isdir("/root/reference/src"):
        import __graft_entry__ as g
        import engines
        engines.build_sim()     # the emulator build of the engine, which fastp_ref_gpusim links
        g.build()               # libfastp_gpu.so, fastp_ref, fastp_ref_gpu (+ fastp_ref_gpusim now that the emulator exists)
        subprocess.check_call([os.path.join(ROOT, "oracle", "build_ref_gpu.sh")], stdout=subprocess.DEVNULL)
        _BUILT = True
    return os.path.exists(REF)


def _run(binary, tmp, tag, flags, paired, gpu_env, threads=1):
    out = os.path.join(tmp, tag)
    os.makedirs(out, exist_ok=True)
    cmd = [binary, "-i", os.path.join(tmp, "in1.fq"), "-o", os.path.join(out, "o1.fq"), "-j", os.path.join(out, "r.json"),
           "-h", os.path.join(out, "r.html"), "-w", str(threads), "--failed_out", os.path.join(out, "failed.fq")]
    if paired:
        cmd += ["-I", os.path.join(tmp, "in2.fq"), "-O", os.path.join(out, "o2.fq")]
    cmd += [x.replace("@TMP@", out) for x in flags]
    env = dict(os.environ)
    env.pop("FASTP_GPU", None)
    env.update(gpu_env)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1200)
    assert p.returncode == 0, f"{os.path.basename(binary)} failed: {p.stderr.decode()[-1500:]}"
    files = {fn: open(os.path.join(out, fn), "rb").read() for fn in sorted(os.listdir(out)) if fn.endswith(".fq")}
    rep = json.load(open(os.path.join(out, "r.json")))
    rep.pop("command", None)
    rep["__stderr__"] = p.stderr.decode(errors="replace")
    return files, rep


def _diff(x, y, path, out):
    if isinstance(x, dict) and isinstance(y, dict):
        for k in sorted(set(x) | set(y)):
            if k not in x or k not in y:
                out.append(f"{path}/{k}: only on one side")
            else:
                _diff(x[k], y[k], f"{path}/{k}", out)
    elif isinstance(x, list) and isinstance(y, list):
        if len(x) != len(y):
            out.append(f"{path}: {len(x)} vs {len(y)} entries")
        else:
            for i, (p, q) in enumerate(zip(x, y)):
                _diff(p, q, f"{path}[{i}]", out)
    elif x != y:
        out.append(f"{path}: reference {x!r} binding {y!r}")


def _check(name, binary, n, tmp_path, seed, threads=1, extra_env=None, eol=b"\n", trailing=True):
    paired, flags, pf, skw = cases.CASES[name]
    flags = list(flags) + BINDING_CASES[name]
    tmp = str(tmp_path)
    d = synth.synth_pairs(n, L=150, seed=seed, paired=paired, **skw)

    def text(seq, qual, lens, mate):   # line ends as the caller wants them (FastqReader::getLine takes \n, \r\n and \r)
        t = synth.to_fastq(seq, qual, lens, mate).replace(b"\n", eol)
        return t if trailing else t[:len(t) - len(eol)]
    with open(os.path.join(tmp, "in1.fq"), "wb") as f:
        f.write(text(d["seq1"], d["qual1"], d["len1"], 1))
    if paired:
        with open(os.path.join(tmp, "in2.fq"), "wb") as f:
            f.write(text(d["seq2"], d["qual2"], d["len2"], 2))
    for fn, content in cases.FILES.get(name, {}).items():
        for tag in ("ref", "gpu"):
            os.makedirs(os.path.join(tmp, tag), exist_ok=True)
            with open(os.path.join(tmp, tag, fn), "wb") as f:
                f.write(content)
    want_files, want_rep = _run(REF, tmp, "ref", flags, paired, {})
    got_files, got_rep = _run(binary, tmp, "gpu", flags, paired, dict({"FASTP_GPU": "1", "FASTP_GPU_VERBOSE": "1"}, **(extra_env or {})),
                              threads=threads)
    err = got_rep.pop("__stderr__")
    want_rep.pop("__stderr__")
    if "overrep" in name:   # -p: the Evaluator's substring census ran on the device too (fastp_gpu_eval_overrep)
        assert err.count("computeOverRepSeq on the device") == (2 if paired else 1), err[-800:]
        if n >= 10000:   # (600 reads do not reach the count thresholds: both sides then agree on "none")
            assert len(want_rep["read1_before_filtering"]["overrepresented_sequences"]) > 0
    assert sorted(want_files) == sorted(got_files)
    for fn in want_files:
        assert want_files[fn] == got_files[fn], f"{name}: {fn} differs ({len(want_files[fn])} vs {len(got_files[fn])} bytes)"
    problems = []
    _diff(want_rep, got_rep, "", problems)
    assert not problems, f"{name}: fastp's own JSON report differs:\n" + "\n".join(problems[:25])
    assert want_rep["summary"]["before_filtering"]["total_reads"] == (2 if paired else 1) * n


# on the emulator (CPU suite) a representative half of the flag sets - every one of them runs against the real library in
# test_gpu_patched_reference_equals_reference; each costs seconds here because the emulator clears Duplicate's 1 GiB
EMULATOR_CASES = ["pe_default", "pe_correction", "pe_merge", "pe_filters", "pe_noadapter_dedup", "pe_umi_per_read",
                  "pe_adapter_fasta", "pe_overrep", "pe_allow_gap_indel", "pe_overlapped_out_trims", "pe_adapter_long",
                  "se_adapter_cut", "se_overrep", "se_adapter_long_indel"]
assert all(n in BINDING_CASES for n in EMULATOR_CASES)


@pytest.mark.parametrize("name", EMULATOR_CASES)
def test_patched_reference_on_emulator_equals_reference(name, tmp_path):
    if not _ensure_built() or not os.path.exists(REF_SIM):
        pytest.skip("reference binaries not built (no /root/reference here)")
    _check(name, REF_SIM, 600, tmp_path, seed=41)


def _auto_adapter_check(binary, n, tmp_path):
    """SE run with adapter auto-detection on reads that carry an adapter fastp does not know (so checkKnownAdapters does
    not short-cut the k-mer path): the Evaluator's ten-mer histogram (evaluator.cpp:384-396) comes from
    fastp_gpu_eval_adapter_kmers; it is compared bin by bin with the reference's own counting loop inside the binding
    (FASTP_GPU_EVAL_CHECK), and the whole run that follows must equal the unpatched reference's"""
    tmp = str(tmp_path)
    custom = b"CTGACCTAGTCAAGGTCCATGCTAGGATCCATGCAAT"
    old = synth.ADAPTER_R1
    synth.ADAPTER_R1 = custom
    try:
        d = synth.synth_pairs(n, L=150, seed=47, paired=False, insert_mean=110.0, insert_sd=25.0, dup_frac=0.0)
    finally:
        synth.ADAPTER_R1 = old
    with open(os.path.join(tmp, "in1.fq"), "wb") as f:
        f.write(synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1))
    flags = ["-G"]
    want_files, want_rep = _run(REF, tmp, "ref", flags, False, {})
    got_files, got_rep = _run(binary, tmp, "gpu", flags, False, {"FASTP_GPU": "1", "FASTP_GPU_VERBOSE": "1", "FASTP_GPU_EVAL_CHECK": "1"})
    err = got_rep.pop("__stderr__")
    want_rep.pop("__stderr__")
    assert "ten-mer histogram on the device" in err, err[-600:]
    # FASTP_GPU_EVAL_CHECK: the binding also ran the reference's own counting loop (Evaluator::seq2int) and compared all 4^10 bins
    import re
    m = re.search(r"ten-mer histogram check vs Evaluator::seq2int: (\d+) of 1048576 bins differ \((\d+) non-zero\)", err)
    assert m and int(m.group(1)) == 0 and int(m.group(2)) > 1000, err[-600:]
    for fn in want_files:
        assert want_files[fn] == got_files[fn], f"{fn} differs"
    problems = []
    _diff(want_rep, got_rep, "", problems)
    assert not problems, "fastp's own JSON report differs:\n" + "\n".join(problems[:25])


def test_patched_reference_auto_adapter_on_emulator(tmp_path):
    if not _ensure_built() or not os.path.exists(REF_SIM):
        pytest.skip("reference binaries not built (no /root/reference here)")
    _auto_adapter_check(REF_SIM, 11000, tmp_path)


@pytest.mark.gpu
def test_gpu_patched_reference_auto_adapter(tmp_path):
    if not (os.path.exists(REF) and os.path.exists(REF_GPU)):
        pytest.skip("oracle/_ref binaries did not travel to this box")
    _auto_adapter_check(REF_GPU, 60000, tmp_path)


@pytest.mark.parametrize("name,threads,packs", [("pe_default", 3, 2), ("se_default_noadapter", 2, 1), ("pe_correction", 2, 3)])
def test_patched_reference_pipelines_windows_of_packs(name, threads, packs, tmp_path):
    """several worker threads, windows of FASTP_GPU_PACKS packs, more windows than slots in flight: the binding packs
    the threads' packs into windows in STREAM order, so the outputs and the whole report equal `fastp_ref -w 1`
    whatever the thread count (duplicates and insert sizes included)"""
    if not _ensure_built() or not os.path.exists(REF_SIM):
        pytest.skip("reference binaries not built (no /root/reference here)")
    _check(name, REF_SIM, 9300, tmp_path, seed=43, threads=threads, extra_env={"FASTP_GPU_PACKS": str(packs)})


@pytest.mark.parametrize("eol,trailing", [(b"\r\n", True), (b"\r", True), (b"\n", False), (b"\r\n", False)])
def test_patched_reference_reader_hook_line_ends(eol, trailing, tmp_path):
    """the memchr hook in front of FastqReader::getLine's scan (fastp_gpu_reader_scan_eol): the same records from \\r\\n, \\r
    and unterminated last lines as the reference's own character-by-character scan (whose binary runs without the hook)"""
    if not _ensure_built() or not os.path.exists(REF_SIM):
        pytest.skip("reference binaries not built (no /root/reference here)")
    _check("pe_cut_right", REF_SIM, 2300, tmp_path, seed=47, threads=2, eol=eol, trailing=trailing)


def test_patched_reference_reader_hook_across_buffer_refills(tmp_path):
    """input files larger than FastqReader's 8 MiB buffer: lines that straddle a refill take getLine's second scan"""
    if not _ensure_built() or not os.path.exists(REF_SIM):
        pytest.skip("reference binaries not built (no /root/reference here)")
    _check("se_default_noadapter", REF_SIM, 52000, tmp_path, seed=48, threads=2, eol=b"\r\n")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(BINDING_CASES))
def test_gpu_patched_reference_equals_reference(name, tmp_path):
    """the same on the real library: fastp_ref_gpu (FASTP_GPU=1) vs fastp_ref, 30 000 units"""
    if not (os.path.exists(REF) and os.path.exists(REF_GPU)):
        pytest.skip("oracle/_ref binaries did not travel to this box")
    _check(name, REF_GPU, 30000, tmp_path, seed=43)
