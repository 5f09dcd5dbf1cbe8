#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference binary.

Run in the build container (needs oracle/_ref/fastp_ref, built from
/root/reference by oracle/build_ref.sh):

    python tests/golden/make_golden.py

For every case of tests/cases.py it writes tests/golden/<case>.npz holding
  * the input FASTQ text (R1[,R2]) - synthetic (tests/synth.py, fixed seed), and
    for case "testdata_pe" the reference's own testdata/R1.fq + R2.fq,
  * md5 + size of every FASTQ the reference wrote (-w 1: out1,out2,failed,merged,overlapped),
  * the reference's JSON report with "command" removed.
The GPU box has no /root/reference; the parity tests read only these files.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
import driver  # noqa: E402
import synth  # noqa: E402

N_PAIRS = 500


def one(name, flags, fq1, fq2):
    ref = driver.run_reference(flags, fq1, fq2, extra_files=cases.FILES.get(name))
    rec = {"fq1": np.frombuffer(fq1, dtype=np.uint8)}
    if fq2 is not None:
        rec["fq2"] = np.frombuffer(fq2, dtype=np.uint8)
    meta = {"flags": flags, "outputs": {}}
    for k in ("out1", "out2", "failed", "merged", "overlapped"):
        b = ref.get(k)
        if b is not None:
            meta["outputs"][k] = {"md5": hashlib.md5(b).hexdigest(), "size": len(b)}
    meta["json"] = ref["json"]
    rec["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(name, {k: v["size"] for k, v in meta["outputs"].items()})


def main():
    only = set(sys.argv[1:])
    for name, (paired, flags, pf, skw) in cases.CASES.items():
        if only and name not in only:
            continue
        d = synth.synth_pairs(cases.N_PAIRS_OVERRIDE.get(name, N_PAIRS), L=150, seed=1234, paired=paired, **skw)
        fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
        fq2 = synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2) if paired else None
        one(name, flags, fq1, fq2)
    td = "/root/reference/testdata"
    if os.path.exists(td) and (not only or "testdata_pe" in only):
        one("testdata_pe", [], open(os.path.join(td, "R1.fq"), "rb").read(),
            open(os.path.join(td, "R2.fq"), "rb").read())


if __name__ == "__main__":
    main()
