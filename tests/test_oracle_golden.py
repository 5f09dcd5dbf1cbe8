"""The CPU oracle against the golden vectors produced by the REAL reference binary
(tests/golden/*.npz) - pins the restatement (trimmed FASTQ, failed_out, merged
output and every JSON number)."""
import pytest

import driver
import golden_util
import oraclelib


@pytest.mark.parametrize("name", golden_util.names())
def test_oracle_matches_reference_golden(name):
    fq1, fq2, meta = golden_util.load(name)
    params = golden_util.params_for(name, fq1=fq1, fq2=fq2)
    eng = oraclelib.Oracle(params)
    try:
        outs, ctr, rep = driver.run_engine(eng, params, fq1, fq2, umi=golden_util.umi_for(name))
    finally:
        eng.close()
    golden_util.check_against_golden(name, outs, rep, meta)


def test_pack_size_does_not_matter():
    """results are a function of the read stream, not of how it is cut into packs"""
    fq1, fq2, meta = golden_util.load("pe_correction")
    params = golden_util.params_for("pe_correction")
    reps = []
    for pack in (1, 7, 1000):
        eng = oraclelib.Oracle(params)
        outs, ctr, rep = driver.run_engine(eng, params, fq1, fq2, pack=pack)
        eng.close()
        reps.append((bytes(outs.out1), bytes(outs.out2), ctr.tobytes()))
    assert reps[0] == reps[1] == reps[2]


@pytest.mark.parametrize("name", golden_util.names())
def test_cpp_host_glue_matches_reference_golden(name):
    """the C++ string side (include/fastp_gpu_host.h) fed with the oracle's records reproduces what the
    reference wrote: same check as above with fastp_amd/hostloop.py swapped for fq_glue.cpp"""
    import engines
    from fastp_amd import engine as eng_mod
    lib = eng_mod.load_library(engines.build_sim())   # host code only: no device needed
    fq1, fq2, meta = golden_util.load(name)
    params = golden_util.params_for(name, fq1=fq1, fq2=fq2)
    eng = oraclelib.Oracle(params)
    try:
        outs, ctr, rep = driver.run_engine(eng, params, fq1, fq2, umi=golden_util.umi_for(name), cpp_host_lib=lib)
    finally:
        eng.close()
    golden_util.check_against_golden(name, outs, rep, meta)
