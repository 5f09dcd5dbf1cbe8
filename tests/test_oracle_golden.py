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
