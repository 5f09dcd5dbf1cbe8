"""Run an engine (CPU oracle or the GPU library) over FASTQ text the way a patched
fastp worker would, and run the real reference binary on the same text."""
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np

from fastp_amd import abi, hostloop
import refjson

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FASTP_REF = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")


def have_reference_binary():
    return os.path.exists(FASTP_REF) and os.access(FASTP_REF, os.X_OK)


def run_engine(engine, params, fq1: bytes, fq2: bytes | None, pack=1000, want_failed=True,
               want_unpaired=False, stride=None, umi=None, cpp_host_lib=None):
    """engine: object with .process(ASCII arrays...) / .counters() / .layout.
    cpp_host_lib: run the string side through the C++ glue (include/fastp_gpu_host.h) of that
    library instead of fastp_amd/hostloop.py"""
    if cpp_host_lib is not None:
        return _run_engine_cpp(engine, params, fq1, fq2, pack, want_failed, want_unpaired, stride, umi, cpp_host_lib)
    b1 = hostloop.parse_fastq(fq1, stride)
    b2 = hostloop.parse_fastq(fq2, b1.seq.shape[1] if stride is None else stride) if fq2 is not None else None
    if b2 is not None and b2.seq.shape[1] != b1.seq.shape[1]:
        st = max(b1.seq.shape[1], b2.seq.shape[1])
        b1 = hostloop.parse_fastq(fq1, st)
        b2 = hostloop.parse_fastq(fq2, st)
    outs = hostloop.Outputs(b2 is not None, want_failed, want_unpaired, want_unpaired)
    amaps = hostloop.AdapterMaps()
    n = b1.n if b2 is None else min(b1.n, b2.n)
    for a in range(0, n, pack):
        e = min(n, a + pack)
        p1 = b1.slice(a, e)
        if b2 is not None:
            p2 = b2.slice(a, e)
            r1, r2, pr, corr = engine.process(p1.seq, p1.qual, p1.lens, p2.seq, p2.qual, p2.lens)
            hostloop.apply_results(params, p1, p2, r1, r2, pr, corr, outs, amaps, umi,
                                   adapter_events=getattr(engine, "last_adapter_events", None))
        else:
            r1, _, _, corr = engine.process(p1.seq, p1.qual, p1.lens)
            hostloop.apply_results(params, p1, None, r1, None, None, corr, outs, amaps, umi,
                                   adapter_events=getattr(engine, "last_adapter_events", None))
    ctr = engine.counters()
    rep = refjson.build(ctr, engine.layout, params, amaps)
    return outs, ctr, rep


def run_reference(flags, fq1: bytes, fq2: bytes | None, want_failed=True, workdir=None, extra_files=None):
    """fastp_ref -w 1 on the same text.  Returns dict(out1,out2,failed: bytes, json: dict)."""
    tmp = workdir or tempfile.mkdtemp(prefix="fastp_ref_")
    for fn, content in (extra_files or {}).items():
        with open(os.path.join(tmp, fn), "wb") as f:
            f.write(content)
    i1 = os.path.join(tmp, "in1.fq")
    with open(i1, "wb") as f:
        f.write(fq1)
    cmd = [FASTP_REF, "-i", i1, "-o", os.path.join(tmp, "o1.fq"), "-j", os.path.join(tmp, "r.json"),
           "-h", os.path.join(tmp, "r.html"), "-w", "1"]
    if fq2 is not None:
        i2 = os.path.join(tmp, "in2.fq")
        with open(i2, "wb") as f:
            f.write(fq2)
        cmd += ["-I", i2, "-O", os.path.join(tmp, "o2.fq")]
    if want_failed:
        cmd += ["--failed_out", os.path.join(tmp, "failed.fq")]
    cmd += [x.replace("@TMP@", tmp) for x in flags]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if p.returncode != 0:
        raise RuntimeError("fastp_ref failed: " + p.stderr.decode()[-2000:])
    res = {}
    for k, fn in (("out1", "o1.fq"), ("out2", "o2.fq"), ("failed", "failed.fq"), ("merged", "merged.fq"),
                  ("overlapped", "overlapped.fq")):
        path = os.path.join(tmp, fn)
        res[k] = open(path, "rb").read() if os.path.exists(path) else None
    res["json"] = refjson.load_reference_json(os.path.join(tmp, "r.json"))
    res["stderr"] = p.stderr.decode()
    return res


def _run_engine_cpp(engine, params, fq1, fq2, pack, want_failed, want_unpaired, stride, umi, lib):
    import cpphost
    b1 = hostloop.parse_fastq(fq1, stride)
    b2 = hostloop.parse_fastq(fq2, b1.seq.shape[1] if stride is None else stride) if fq2 is not None else None
    if b2 is not None and b2.seq.shape[1] != b1.seq.shape[1]:
        st = max(b1.seq.shape[1], b2.seq.shape[1])
        b1 = hostloop.parse_fastq(fq1, st)
        b2 = hostloop.parse_fastq(fq2, st)
    host = cpphost.CppHost(lib, params, want_failed, want_unpaired, umi)
    n = b1.n if b2 is None else min(b1.n, b2.n)
    for a in range(0, n, pack):
        e = min(n, a + pack)
        p1 = b1.slice(a, e)
        p2 = b2.slice(a, e) if b2 is not None else None
        if p2 is not None:
            r1, r2, pr, corr = engine.process(p1.seq, p1.qual, p1.lens, p2.seq, p2.qual, p2.lens)
        else:
            r1, r2, pr, corr = engine.process(p1.seq, p1.qual, p1.lens)
        host.apply(p1, p2, r1, r2, pr, corr, getattr(engine, "last_adapter_events", None))
    outs = host.outputs(b2 is not None)
    if not want_failed:
        outs.failed = None
    if not want_unpaired:
        outs.unpaired1 = outs.unpaired2 = None
    amaps = host.adapter_maps()
    host.close()
    ctr = engine.counters()
    rep = refjson.build(ctr, engine.layout, params, amaps)
    return outs, ctr, rep


def md5(b):
    return hashlib.md5(b if b is not None else b"").hexdigest()
