"""Oracle vs the real reference binary, run live (only where oracle/_ref/fastp_ref
exists, i.e. in the build container / on a box the prebuilt binary travelled to)."""
import pytest

import cases
import driver
import oraclelib
import refjson
import synth
from fastp_amd import hostloop

pytestmark = pytest.mark.skipif(not driver.have_reference_binary(), reason="oracle/_ref/fastp_ref not built")


@pytest.mark.parametrize("name", ["pe_cut_right", "pe_adapter_seq", "pe_correction", "se_adapter_cut",
                                  "pe_polyg_polyx", "pe_adapter_fasta", "se_adapter_fasta", "pe_overrep", "se_overrep"])
def test_oracle_equals_reference_live(name, tmp_path):
    paired, flags, pf, skw = cases.CASES[name]
    d = synth.synth_pairs(4000, L=100, seed=99, paired=paired, **skw)   # a different length/seed than golden
    fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
    fq2 = synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2) if paired else None
    params = cases.finalize_params(name, pf(104), d["seq1"], d["len1"], d.get("seq2"), d.get("len2"))
    ref = driver.run_reference(flags, fq1, fq2, workdir=str(tmp_path), extra_files=cases.FILES.get(name))
    eng = oraclelib.Oracle(params)
    umi = hostloop.UmiNameEditor(*cases.UMI[name]) if name in cases.UMI else None
    outs, ctr, rep = driver.run_engine(eng, params, fq1, fq2, umi=umi)
    eng.close()
    assert (ref["out1"] or b"") == bytes(outs.out1)
    if paired:
        assert (ref["out2"] or b"") == bytes(outs.out2)
    assert (ref["failed"] or b"") == bytes(outs.failed)
    assert refjson.diff(ref["json"], rep) == []
