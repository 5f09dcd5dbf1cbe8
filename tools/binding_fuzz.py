#!/usr/bin/env python3
"""Differential fuzz of the WHOLE drop-in: random fastp command lines x random synthetic inputs through the reference
(oracle/_ref/fastp_ref) and through the patched reference (FASTP_GPU=1: fastp_ref_gpu on a GPU box, fastp_ref_gpusim - the
same binding on the SIMT emulator - elsewhere); every output file and the reference's own JSON report must be equal.
What the engine-level fuzz (tests/test_option_fuzz.py) cannot see is covered here: the command line -> parameter block
mapping of the binding (oracle/patches/gpu_worker.cpp fill_params), the device parser / formatter, the adapter replay.

    python tools/binding_fuzz.py FIRST LAST [--gpu]      one line per failing seed, a summary at the end
"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
import test_ref_binding as rb  # noqa: E402

ADAPTER_R1 = "AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
ADAPTER_R2 = "AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"


def random_command(seed):
    rng = np.random.default_rng(seed)
    pick = lambda p: bool(rng.random() < p)   # noqa: E731
    paired = pick(0.7)
    f = []
    L = int(rng.choice([75, 100, 150, 150, 151]))
    # adapters
    if pick(0.2):
        f += ["-A"]
    elif pick(0.4):
        f += ["-a", ADAPTER_R1]
        if paired and pick(0.7):
            f += ["--adapter_sequence_r2", ADAPTER_R2]
    if paired and "-A" not in f and "-a" not in f and pick(0.3):
        f += ["--detect_adapter_for_pe"]   # the Evaluator's adapter detection on both mates (its counting loop runs on the device)
    if paired and pick(0.15):
        f += ["--allow_gap_overlap_trimming"]
    if pick(0.15):
        f += ["--dimer_max_len", str(int(rng.integers(0, 40)))]
    # fixed trims
    for flag, prob, hi in (("-f", 0.2, 12), ("-t", 0.2, 12), ("-b", 0.15, None), ("-F", 0.15, 12), ("-T", 0.15, 12), ("-B", 0.1, None)):
        if flag in ("-F", "-T", "-B") and not paired:
            continue
        if pick(prob):
            f += [flag, str(int(rng.integers(L // 2, L)) if hi is None else int(rng.integers(1, hi)))]
    # duplication
    if pick(0.15):
        f += ["--dont_eval_duplication"]
    else:
        if pick(0.25):
            f += ["--dedup"]
        if pick(0.2):
            f += ["--dup_calc_accuracy", str(int(rng.choice([1, 2, 3])))]
    # polyG / polyX
    f += ["-G"] if pick(0.5) else (["-g"] + (["--poly_g_min_len", str(int(rng.integers(5, 20)))] if pick(0.4) else []))
    if pick(0.25):
        f += ["-x"] + (["--poly_x_min_len", str(int(rng.integers(5, 20)))] if pick(0.4) else [])
    # quality cutting
    cut = [c for c in ("--cut_front", "--cut_tail", "--cut_right") if pick(0.3)]
    f += cut
    if cut and pick(0.4):
        f += ["-W", str(int(rng.integers(1, 9))), "-M", str(int(rng.integers(5, 31)))]
    for c in cut:
        if pick(0.3):
            f += [c + "_window_size", str(int(rng.integers(1, 9)))]
        if pick(0.3):
            f += [c + "_mean_quality", str(int(rng.integers(5, 31)))]
    # filters
    if pick(0.15):
        f += ["-Q"]
    else:
        if pick(0.3):
            f += ["-q", str(int(rng.integers(5, 31)))]
        if pick(0.3):
            f += ["-u", str(int(rng.integers(5, 80)))]
        if pick(0.3):
            f += ["-n", str(int(rng.integers(0, 8)))]
        if pick(0.25):
            f += ["-e", str(int(rng.integers(10, 32)))]
    if pick(0.15):
        f += ["-L"]
    else:
        if pick(0.3):
            f += ["-l", str(int(rng.integers(10, 80)))]
        if pick(0.15):
            f += ["--length_limit", str(int(rng.integers(L - 40, L)))]
    if pick(0.2):
        f += ["-y"] + (["-Y", str(int(rng.integers(10, 60)))] if pick(0.5) else [])
    # paired-end analyses
    umi = False
    if paired:
        if pick(0.3):
            f += ["-c"]
        if pick(0.25):
            f += ["-m", "--merged_out", "@TMP@/merged.fq"] + (["--include_unmerged"] if pick(0.5) else [])
        if pick(0.15):
            f += ["--overlapped_out", "@TMP@/overlapped.fq"]
        if pick(0.2):
            f += ["--overlap_len_require", str(int(rng.integers(10, 40)))]
        if pick(0.2):
            f += ["--overlap_diff_limit", str(int(rng.integers(1, 10)))]
        if pick(0.2):
            f += ["--overlap_diff_percent_limit", str(int(rng.integers(5, 40)))]
        if pick(0.15) and "-m" not in f:
            f += ["--unpaired1", "@TMP@/u1.fq", "--unpaired2", "@TMP@/u2.fq"]
    if pick(0.15):
        umi = True
        loc = str(rng.choice(["read1", "read2", "per_read"] if paired else ["read1"]))
        f += ["-U", "--umi_loc", loc, "--umi_len", str(int(rng.integers(3, 10)))]
        if pick(0.4):
            f += ["--umi_skip", str(int(rng.integers(1, 4)))]
        if pick(0.3):
            f += ["--umi_prefix", "UMI"]
        if pick(0.2):
            f += ["--umi_delim", "_"]
    if pick(0.2):
        f += ["-p", "-P", str(int(rng.choice([1, 2, 5, 20])))]
    if pick(0.15):
        f += ["--reads_to_process", str(int(rng.integers(100, 1200)))]
    n = int(rng.integers(300, 1300))
    skw = dict(insert_mean=float(rng.choice([0.9, 1.3, 1.8])) * L, insert_sd=0.4 * L, polyg_frac=float(rng.choice([0.0, 0.15])),
               polyx_frac=float(rng.choice([0.0, 0.2])), dup_frac=float(rng.choice([0.05, 0.3])), ragged_frac=float(rng.choice([0.0, 0.05, 0.3])),
               lowq_site_rate=float(rng.choice([0.01, 0.05])), exotic_frac=float(rng.choice([0.0, 0.0, 0.05, 0.3])))
    eol = [b"\n", b"\n", b"\r\n", b"\r"][int(rng.integers(0, 4))]
    threads = int(rng.choice([1, 2, 5]))
    gz = pick(0.15)
    mode = "pack" if pick(0.2) else "stream"
    # ".gz" inputs, drawn from a generator of their own so that a seed's command line stays what it was before they existed:
    # bgzip-written (inflated on the device), one gzip member, several members (zlib inside the stream); pack mode leaves them
    # to the reference's reader
    rng2 = np.random.default_rng(seed + 7_000_000)
    gz_in = None
    if rng2.random() < 0.3:
        kinds = ["bgzf", "bgzf", "gzip", "members"]
        gz_in = tuple(kinds[int(rng2.integers(0, 4))] for _ in range(2 if paired else 1))
    interleaved = bool(paired and rng2.random() < 0.2)   # --interleaved_in: both mates in one file
    if interleaved and gz_in:
        gz_in = gz_in[:1]
    if interleaved and "--detect_adapter_for_pe" in f:   # the reference itself cannot: its Evaluator opens <in2> ("Failed to open file: ")
        f = [x for x in f if x != "--detect_adapter_for_pe"]
    phred64 = bool(rng2.random() < 0.15)                 # --phred64: the input's qualities on the +64 scale
    if phred64:
        f = f + ["--phred64"]
    return dict(paired=paired, flags=f, L=L, n=n, skw=skw, eol=eol, threads=threads, gz=gz, mode=mode, umi=umi, gz_in=gz_in, interleaved=interleaved,
                phred64=phred64)


def run(seed, binary, sim):
    c = random_command(seed)
    tmp = tempfile.mkdtemp(prefix="bfz")
    try:
        d = synth.synth_pairs(c["n"], L=c["L"], seed=seed, paired=c["paired"], **c["skw"])
        if c["phred64"]:
            rb._to_phred64(d)
        open(os.path.join(tmp, "in1.fq"), "wb").write(synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1).replace(b"\n", c["eol"]))
        if c["paired"]:
            open(os.path.join(tmp, "in2.fq"), "wb").write(synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2).replace(b"\n", c["eol"]))
        if c["interleaved"]:
            a = open(os.path.join(tmp, "in1.fq"), "rb").read().split(c["eol"])
            b = open(os.path.join(tmp, "in2.fq"), "rb").read().split(c["eol"])
            recs = []
            for i in range(0, min(len(a), len(b)) - 3, 4):
                recs += a[i:i + 4] + b[i:i + 4]
            open(os.path.join(tmp, "in1.fq"), "wb").write(c["eol"].join(recs) + c["eol"])
            os.remove(os.path.join(tmp, "in2.fq"))
        in1 = in2 = None
        if c["gz_in"]:
            in1, in2 = rb._compress_inputs(tmp, c["paired"], c["gz_in"])
        want_files, want_rep = rb._run(rb.REF, tmp, "ref", c["flags"], c["paired"], {}, gz=c["gz"], in1=in1, in2=in2, interleaved=c["interleaved"])
        env = {"FASTP_GPU": "1"}
        if sim:
            env.update(rb.SIM_ENV)
        if c["mode"] == "pack":
            env.update(rb.PACK_MODE)
        got_files, got_rep = rb._run(binary, tmp, "gpu", c["flags"], c["paired"], env, threads=c["threads"], gz=c["gz"], in1=in1, in2=in2, interleaved=c["interleaved"])
        want_rep.pop("__stderr__")
        got_rep.pop("__stderr__")
        problems = []
        if sorted(want_files) != sorted(got_files):
            problems.append(f"files {sorted(want_files)} vs {sorted(got_files)}")
        for k in want_files:
            if k in got_files and want_files[k] != got_files[k]:
                problems.append(f"{k} differs ({len(want_files[k])} vs {len(got_files[k])} bytes)")
        rb._diff(want_rep, got_rep, "", problems)
        return problems, c
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    first, last = int(sys.argv[1]), int(sys.argv[2])
    gpu = "--gpu" in sys.argv
    binary = rb.REF_GPU if gpu else rb.REF_SIM
    t0 = time.time()
    ok = failed = 0
    for seed in range(first, last):
        try:
            problems, c = run(seed, binary, not gpu)
        except AssertionError as e:   # a binary refused the command line / failed
            problems, c = [f"run failed: {str(e)[-400:]}"], random_command(seed)
        if problems:
            failed += 1
            print(f"seed {seed}: {' '.join(c['flags'])} [{c['mode']}, -w {c['threads']},  paired={c['paired']}, gz={c['gz']}, gz_in={c['gz_in']}, interleaved={c['interleaved']}, phred64={c['phred64']}]: " + "; ".join(problems[:4]), flush=True)
        else:
            ok += 1
    print(f"seeds {first}..{last - 1}: {ok} command lines with every output file and the JSON report equal to the reference's, {failed} FAILED, {time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
