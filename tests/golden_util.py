"""Load the committed golden fixtures (tests/golden/*.npz, made by make_golden.py)."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(z["meta"].tobytes().decode())
    fq1 = z["fq1"].tobytes()
    fq2 = z["fq2"].tobytes() if "fq2" in z.files else None
    return fq1, fq2, meta


def params_for(name, max_len=152, fq1=None, fq2=None):
    import cases
    from fastp_amd import abi, hostloop
    if name == "testdata_pe":
        p = abi.default_params(True, max_len)
        p.poly_g = 1  # reference auto-enables it: read names start with @A (evaluator.cpp:16-45)
        return p
    p = cases.CASES[name][2](max_len)
    if name in cases.OVERREP:  # the Evaluator pre-pass over the input (host logic)
        b1 = hostloop.parse_fastq(fq1)
        b2 = hostloop.parse_fastq(fq2) if fq2 is not None else None
        p = cases.finalize_params(name, p, b1.seq, b1.lens, b2.seq if b2 else None, b2.lens if b2 else None)
    return p


def umi_for(name):
    import cases
    from fastp_amd import hostloop
    return hostloop.UmiNameEditor(*cases.UMI[name]) if name in cases.UMI else None


def check_against_golden(name, outs, rep, meta):
    """assert engine outputs/report equal what the reference wrote"""
    import refjson
    from driver import md5
    problems = []
    for k, exp in meta["outputs"].items():
        got = bytes(getattr(outs, k if k != "out2" else "out2") or b"")
        if md5(got) != exp["md5"]:
            problems.append(f"{k}: md5 {md5(got)} size {len(got)} != reference {exp['md5']} size {exp['size']}")
    problems += refjson.diff(meta["json"], rep)
    assert not problems, f"{name}: " + "\n".join(problems[:30])
