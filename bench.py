#!/usr/bin/env python3
"""bench.py - Mreads/s of the fused per-read path on synthetic 2x150 bp paired-end batches
resident in HBM (BASELINE.json metric; workload = configs[2]: PE 2x150, auto-adapter via
overlap + quality-trim), with the kernel's roofline position and the reference's CPU path
timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs B]

A step = one pass of the hot path (fused kernel + slab fold + duplicate kernels) over one
batch of B pairs already in HBM.  N > 1: launched by torch.distributed.run, one rank per GPU,
every rank owns its own batch (weak scaling, no data-path collective) and the counter blocks
are merged by one RCCL all-reduce at the end, inside the timed region.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy
# the kernels whose summed duration per launch is `roofline.kernel_avg_ms` (one HIP-event pair around them)
KERNELS = {"fused": "fq_fused_kernel",
           "split": "fq_scan_kernel + fq_stats_kernel (the fused kernel's work as two launches)",
           "lane": "fq_lane_kernel + fq_stats5_kernel (the fused kernel's work as two launches; form 5 of the Stats kernel at this read length)"}
L = 150


def algorithmic_bytes_per_pair(L):
    # per read ceil(L/4) packed bases + L quality + 4 length in, 16 result out
    return 2 * ((L + 3) // 4 + L + 4) + 2 * 16   # 416 at L=150


def bench_params():
    from fastp_amd import abi
    p = abi.default_params(True, L)      # adapter trimming by overlap, dup evaluation, filters: fastp defaults
    p.cut_right = 1                      # + sliding-window quality trim (configs[2])
    return p, ["-G", "--cut_right"]


def write_sample_files(sample_pairs, dev):
    """the bounded sample of the workload as plain FASTQ on tmpfs (generated on the GPU)"""
    import torch
    import synth_torch
    need = sample_pairs * 4 * (2 * L + 60) * 2
    base = None
    for cand in ("/dev/shm", "/tmp"):
        try:
            st = os.statvfs(cand)
            if st.f_bavail * st.f_frsize > need:
                base = cand
                break
        except OSError:
            pass
    tmp = tempfile.mkdtemp(prefix="fastp_cpu_", dir=base)
    f1, f2 = os.path.join(tmp, "r1.fq"), os.path.join(tmp, "r2.fq")
    block = 1_000_000
    with open(f1, "wb", buffering=0) as a, open(f2, "wb", buffering=0) as b:
        for done in range(0, sample_pairs, block):
            k = min(block, sample_pairs - done)
            d = synth_torch.synth_pairs_torch(k, L=L, seed=4242 + done // block, device=dev)
            for mate, fh in ((1, a), (2, b)):
                rec = synth_torch.to_fastq_tensor(d[f"seq{mate}"], d[f"qual{mate}"], mate, first=done).cpu().numpy()
                fh.write(memoryview(rec).cast("B"))
            del d
    torch.cuda.empty_cache()
    return tmp, f1, f2


def host_cores():
    return min(os.cpu_count() or 1, 16)   # the reference stops scaling long before that (reader-thread bound)


def fresh_outputs(cmd):
    """every timed run writes NEW output files: opening an existing file of gigabytes with O_TRUNC frees its tmpfs pages first, 0.1 - 0.5 s
    that belong to the previous run (profiles/r06_r_dropin_probe.txt: the runs behind the first of a series were that much slower)"""
    for i, a in enumerate(cmd):
        if a in ("-o", "-O") and os.path.exists(cmd[i + 1]):
            os.remove(cmd[i + 1])


def timed_ref(binary, tmp, f1, f2, flags, cores, env=None, tag="o"):
    cmd = [binary, "-i", f1, "-I", f2, "-o", os.path.join(tmp, tag + "1.fq"), "-O", os.path.join(tmp, tag + "2.fq"),
           "-j", os.path.join(tmp, tag + ".json"), "-h", os.path.join(tmp, tag + ".html"), "-w", str(cores)] + flags
    times = []
    for _ in range(3):   # median of 3: the reference's own convention (scripts/bench_e2e.sh:10), SURVEY.md 8(d)
        fresh_outputs(cmd)
        t0 = time.time()
        subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=900,
                       env=dict(os.environ, **(env or {})))
        times.append(time.time() - t0)
    return sorted(times)[1]


def cpu_baseline(sample_pairs, flags, params, dev, files=None):
    """reference fastp (oracle/_ref/fastp_ref, scalar-SIMD shim build) on the host cores, on a bounded
    sample of the same workload (generated on the GPU, written to tmpfs as plain FASTQ); falls back to the
    plain-C oracle port on a smaller sample if the binary is absent."""
    import numpy as np
    import synth_torch
    ref = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")
    cores = host_cores()
    if os.path.exists(ref) and files is not None:
        tmp, f1, f2 = files
        wall = timed_ref(ref, tmp, f1, f2, flags, cores)
        return {"value": round(2 * sample_pairs / wall / 1e6, 4), "unit": "Mreads/s", "cores": cores,
                "kind": "reference",
                "sample": f"{sample_pairs} synthetic 2x{L} pairs, plain FASTQ -> FASTQ on tmpfs, fastp_ref -w {cores} "
                          f"(scalar shim for Highway SIMD; the box has {os.cpu_count()} logical cores), end-to-end wall incl. "
                          f"FASTQ parse/write and the reference's start-up (bloom allocation, pre-pass), median of 3"}
    import oraclelib
    sample_pairs = min(sample_pairs, 500_000)
    d = synth_torch.synth_pairs_torch(sample_pairs, L=L, seed=4242, device=dev)
    orc = oraclelib.Oracle(params)
    pad = lambda a: np.pad(a.cpu().numpy(), ((0, 0), (0, 2)))
    t0 = time.time()
    orc.process(pad(d["seq1"]), pad(d["qual1"]), d["len1"].cpu().numpy(), pad(d["seq2"]), pad(d["qual2"]), d["len2"].cpu().numpy())
    wall = time.time() - t0
    orc.close()
    return {"value": round(2 * sample_pairs / wall / 1e6, 4), "unit": "Mreads/s", "cores": 1, "kind": "port",
            "sample": f"{sample_pairs} synthetic 2x{L} pairs through the plain-C oracle (per-read loop only, 1 thread)"}


def report_sections_that_differ(path_a, path_b, ignore=("command",)):
    """top-level sections of two fastp JSON reports whose contents differ (scripts/bench_e2e.sh:171-209 compares the outputs; the report
    is what the counter block produces: every Stats / FilterResult / Duplicate number)"""
    a, b = json.load(open(path_a)), json.load(open(path_b))
    return sorted(k for k in set(a) | set(b) if k not in ignore and a.get(k) != b.get(k))


def e2e_legs(sample_pairs, flags, params, files, cpu_value, dropin_only=False, full_json=False):
    """the same files end to end through the GPU path, two ways (never the headline `value`):
    e2e_gpu    : fastp_amd.pipeline - raw text to HBM, parse / worker loop / format on the device, text back, file I/O
    e2e_dropin : the real reference with its worker loops bound to the engine (oracle/_ref/fastp_ref_gpu, FASTP_GPU=1)
                 - the reference's own reader / writer threads around the C ABI - next to fastp_ref on the same files"""
    out = {}
    tmp, f1, f2 = files
    try:
        if dropin_only:
            raise StopIteration
        from fastp_amd import pipeline
        pl = pipeline.FastqPipeline(params, chunk_bytes=256 << 20)
        best = None
        for _ in range(2):
            t0 = time.time()
            pl.run(f1, f2, os.path.join(tmp, "g1.fq"), os.path.join(tmp, "g2.fq"))
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
        pl.close()
        out["e2e_gpu"] = {"value": round(2 * sample_pairs / best / 1e6, 3), "unit": "Mreads/s",
                          "what": "fastp_amd.pipeline FASTQ -> FASTQ on tmpfs, parse + worker loop + format on the device, best of 2"}
    except StopIteration:
        pass
    except Exception as e:   # the kernel line must not depend on the file pipeline
        out["e2e_gpu"] = {"value": None, "error": repr(e)[:200]}
    refgpu = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu")
    if os.path.exists(refgpu):
        cores = host_cores()
        try:
            import re
            cmd = [refgpu, "-i", f1, "-I", f2, "-o", os.path.join(tmp, "d1.fq"), "-O", os.path.join(tmp, "d2.fq"), "-j", os.path.join(tmp, "d.json"),
                   "-h", os.path.join(tmp, "d.html"), "-w", str(cores)] + flags
            best, stream_s, setup_s = None, None, None
            for _ in range(2):
                fresh_outputs(cmd)
                t0 = time.time()
                pr = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True, timeout=900,
                                    env=dict(os.environ, FASTP_GPU="1", FASTP_GPU_VERBOSE="1"))
                dt = time.time() - t0
                if best is None or dt < best:
                    best = dt
                    m = re.search(r"stream mode: \d+ units in \d+ chunks, ([0-9.]+) s \(setup ([0-9.]+)", pr.stderr.decode(errors="replace"))
                    stream_s, setup_s = (float(m.group(1)), float(m.group(2))) if m else (None, None)
            same = None
            o1, d1 = os.path.join(tmp, "o1.fq"), os.path.join(tmp, "d1.fq")
            if os.path.exists(o1) and os.path.exists(d1):
                import hashlib

                def md5(pth):
                    hsh = hashlib.md5()
                    with open(pth, "rb") as fh:
                        for blk in iter(lambda: fh.read(1 << 24), b""):
                            hsh.update(blk)
                    return hsh.hexdigest()
                same = md5(o1) == md5(d1) and md5(os.path.join(tmp, "o2.fq")) == md5(os.path.join(tmp, "d2.fq"))
            # The report: the reference's own JSON depends on -w in exactly two sections (insert_size: sampled on worker thread 0 only,
            # peprocessor.cpp:449,497; adapter_cutting: one FilterResult map per thread with its own caps, filterresult.cpp:38-89 -
            # profiles/r04_ref_thread_dependence.txt), the binding has fastp_ref -w 1's.  Against the -w <cores> report of the CPU
            # leg everything else must be identical; against a -w 1 run of the same files (full_json: the 4 M-pair sample) all of it.
            json_diff_w, json_diff_1 = None, None
            oj, dj = os.path.join(tmp, "o.json"), os.path.join(tmp, "d.json")
            if os.path.exists(oj) and os.path.exists(dj):
                json_diff_w = [k for k in report_sections_that_differ(oj, dj) if k not in ("insert_size", "adapter_cutting")]
                if full_json:
                    ref1 = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")
                    subprocess.run([ref1, "-i", f1, "-I", f2, "-o", os.path.join(tmp, "s1.fq"), "-O", os.path.join(tmp, "s2.fq"), "-j",
                                    os.path.join(tmp, "s.json"), "-h", os.path.join(tmp, "s.html"), "-w", "1"] + flags,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=900)
                    json_diff_1 = report_sections_that_differ(os.path.join(tmp, "s.json"), dj)
                    if same is not None:
                        same = same and md5(os.path.join(tmp, "s1.fq")) == md5(d1)
                    for n in ("s1.fq", "s2.fq"):
                        os.remove(os.path.join(tmp, n))
                if same is not None:   # outputs_identical = the FASTQ files AND the report
                    same = same and not json_diff_w and not json_diff_1
            # what a run costs whatever its size: process start, HIP runtime, engine + page-locked buffers (the first 1000 pairs only)
            fresh_outputs(cmd)
            t0 = time.time()
            subprocess.run(cmd + ["--reads_to_process", "1000"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=300,
                           env=dict(os.environ, FASTP_GPU="1"))
            startup = time.time() - t0
            out["e2e_dropin"] = {"gpu": round(2 * sample_pairs / best / 1e6, 3), "cpu": cpu_value, "unit": "Mreads/s", "cores": cores,
                                 "outputs_identical": same,
                                 "report_sections_differing_from_w%d" % cores: json_diff_w,
                                 "report_sections_differing_from_w1": json_diff_1,
                                 "wall_s": round(best, 3), "startup_s": round(startup, 3),
                                 "stream_s": stream_s, "stream_setup_s": setup_s,
                                 "stream_Mreads_per_s": round(2 * sample_pairs / stream_s / 1e6, 2) if stream_s else None,
                                 "what": f"FASTP_GPU=1 fastp_ref_gpu -w {cores} (stream binding: raw chunks -> device parser / worker loop / formatter -> "
                                         f"the writers' files) vs fastp_ref -w {cores}, same files on tmpfs, whole-process wall, best of 2; startup_s = the same "
                                         f"binary on the first 1000 pairs; stream_s = the file loop alone; outputs_identical = out1 / out2 md5 AND the "
                                         f"JSON report minus `command` (against fastp_ref -w {cores}: minus the two sections the reference itself makes "
                                         f"depend on -w; against fastp_ref -w 1 where report_sections_differing_from_w1 is a list: the whole report)"}
        except Exception as e:
            out["e2e_dropin"] = {"gpu": None, "error": repr(e)[:200]}
    return out


def e2e_compressed_leg(sample_pairs, flags, dev):
    """".gz" on both sides of the drop-in (never the headline `value`): (a) the patched reference writes its outputs as gzip members
    made on the device; (b) those files - bgzip members, the format the reference reads with BgzfMtReader - are the INPUT of a second
    run, which ships them to the device compressed and inflates them there; (c) fastp_ref reads the same ".gz" files on the host
    cores (its reader inflates through oracle/shims/isa-l over zlib here).  Outputs of (b) and (c) are compared."""
    import hashlib
    import json as _json
    import re
    import shutil
    ref = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")
    refgpu = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu")
    if not (os.path.exists(ref) and os.path.exists(refgpu)):
        return None
    cores = host_cores()
    tmp, f1, f2 = write_sample_files(sample_pairs, dev)
    try:
        J = lambda n: os.path.join(tmp, n)
        genv = dict(os.environ, FASTP_GPU="1", FASTP_GPU_VERBOSE="1")

        def run(binary, i1, i2, tag, ext, env):
            cmd = [binary, "-i", i1, "-I", i2, "-o", J(tag + "1" + ext), "-O", J(tag + "2" + ext), "-j", J(tag + ".json"), "-h", J(tag + ".html"),
                   "-w", str(cores)] + flags
            fresh_outputs(cmd)
            t0 = time.time()
            pr = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True, timeout=240, env=env)
            return time.time() - t0, pr.stderr.decode(errors="replace")

        def md5(pth):
            hsh = hashlib.md5()
            with open(pth, "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    hsh.update(blk)
            return hsh.hexdigest()
        t_a, _ = run(refgpu, f1, f2, "z", ".fq.gz", genv)
        kept = _json.load(open(J("z.json")))["summary"]["after_filtering"]["total_reads"]
        # (d) the same inputs as ORDINARY gzip files (no bgzip fields: what sequencers and pigz deliver; eight members per file so
        # that making them takes seconds): the stream inflates each with several host threads (fastp_amd/csrc/fq_pgunzip.h), the
        # reference with its one reader thread per file.  Never at the expense of the legs above and below.
        plain_gz = None
        try:
            t0 = time.time()
            procs, parts = [], {f1: [], f2: []}
            for f in (f1, f2):
                size = os.path.getsize(f)
                step = (size + 7) // 8
                for i in range(8):
                    part = J(os.path.basename(f) + f".part{i}.gz")
                    parts[f].append(part)
                    procs.append(subprocess.Popen(["bash", "-c", f"tail -c +{i * step + 1} '{f}' | head -c {step} | gzip -1 > '{part}'"]))
            if any(p.wait() != 0 for p in procs):
                raise RuntimeError("gzip failed")
            for f, tag in ((f1, "p1.fq.gz"), (f2, "p2.fq.gz")):
                with open(J(tag), "wb") as dst:
                    for part in parts[f]:
                        with open(part, "rb") as src:
                            shutil.copyfileobj(src, dst, 1 << 24)
                        os.remove(part)
            t_gz = time.time() - t0
            best_p, err_p = None, ""
            for _ in range(2):
                t_p, e = run(refgpu, J("p1.fq.gz"), J("p2.fq.gz"), "g", ".fq", genv)
                if best_p is None or t_p < best_p:
                    best_p, err_p = t_p, e
            t_q, _ = run(ref, J("p1.fq.gz"), J("p2.fq.gz"), "h", ".fq", dict(os.environ))
            mp = re.search(r"stream mode: \d+ units in (\d+) chunks, ([0-9.]+) s", err_p)
            hw = os.cpu_count() or 1
            plain_gz = {"gpu": round(kept / best_p / 1e6, 3), "cpu": round(kept / t_q / 1e6, 3), "unit": "Mreads/s",
                        "outputs_identical": md5(J("g1.fq")) == md5(J("h1.fq")) and md5(J("g2.fq")) == md5(J("h2.fq")),
                        "inflated_on_host_threads": "inflated on host threads" in err_p,
                        "inflater_threads_per_file": int(os.environ.get("FASTP_GPU_STREAM_GUNZIP_THREADS", max(1, min(12, hw // 4)))),
                        "compressed_bytes": os.path.getsize(J("p1.fq.gz")) + os.path.getsize(J("p2.fq.gz")),
                        "wall_s": round(best_p, 3), "stream_s": float(mp.group(2)) if mp else None, "making_the_files_s": round(t_gz, 1),
                        "what": f"the same {sample_pairs} pairs as two ordinary gzip files (gzip -1, eight members each): FASTP_GPU=1 fastp_ref_gpu "
                                f"(fq_pgunzip.h: several host threads per stream) vs fastp_ref -w {cores} (one reader thread per file, zlib behind "
                                f"the ISA-L shim), whole-process wall, best of 2"}
            for n in ("p1.fq.gz", "p2.fq.gz", "g1.fq", "g2.fq", "h1.fq", "h2.fq"):
                if os.path.exists(J(n)):
                    os.remove(J(n))
        except Exception as e:   # noqa: BLE001
            plain_gz = {"gpu": None, "error": repr(e)[:300]}
        os.remove(f1)
        os.remove(f2)
        best, err = None, ""
        for _ in range(2):
            t_b, e = run(refgpu, J("z1.fq.gz"), J("z2.fq.gz"), "b", ".fq", genv)
            if best is None or t_b < best:
                best, err = t_b, e
        t_c, _ = run(ref, J("z1.fq.gz"), J("z2.fq.gz"), "c", ".fq", dict(os.environ))
        m = re.search(r"stream mode: \d+ units in (\d+) chunks, ([0-9.]+) s", err)
        k = re.search(r"inflate \+ its copy to the host ([0-9.]+) s", err)
        return {"gz_outputs_Mreads_per_s": round(2 * sample_pairs / t_a / 1e6, 3),
                "gpu": round(kept / best / 1e6, 3), "cpu": round(kept / t_c / 1e6, 3), "unit": "Mreads/s", "cores": cores, "reads": kept,
                "outputs_identical": md5(J("b1.fq")) == md5(J("c1.fq")) and md5(J("b2.fq")) == md5(J("c2.fq")),
                "inflated_on_the_device": "BGZF, inflated on the device" in err,
                "compressed_bytes": os.path.getsize(J("z1.fq.gz")) + os.path.getsize(J("z2.fq.gz")),
                "wall_s": round(best, 3), "stream_s": float(m.group(2)) if m else None, "chunks": int(m.group(1)) if m else None,
                "inflate_s": float(k.group(1)) if k else None, "plain_gzip_inputs": plain_gz,
                "what": f"FASTP_GPU=1 fastp_ref_gpu -w {cores}: {sample_pairs} pairs written as .fq.gz (gzip members made on the device), then those "
                        f"files as the input of a second run (compressed to HBM, fastp_gpu_inflate_bgzf in place of BgzfMtReader) vs fastp_ref -w {cores} "
                        f"on the same .gz files (zlib behind the ISA-L shim), whole-process wall, best of 2"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def other_configs(dev, only=None):
    """the other single-GPU BASELINE.json configurations, inputs resident in HBM (not bench lines: reported beside it);
    only: run the configurations whose name contains this text (several: separated by |)"""
    import numpy as np
    import torch
    import synth_torch
    from fastp_amd import abi, engine
    res = []

    def run(name, params, Lr, n, paired, steps=12, soft_masked_every=0):
        if only and not any(o in name for o in only.split("|")):
            return
        d = synth_torch.synth_pairs_torch(n, L=Lr, seed=5, device=dev)
        bufs = {}
        for m in ("1", "2") if paired else ("1",):
            bufs[m] = synth_torch.pack_torch(d["seq" + m], d["qual" + m], d["len" + m], Lr)
        xkeep = None
        if soft_masked_every:   # every k-th unit in lower case: letters outside ACGTN, the text kernel (fq_text.h) takes those units
            xu = np.arange(0, n, soft_masked_every, dtype=np.int32)
            xt = torch.from_numpy(xu.astype(np.int64)).to(dev)
            xkeep = [xu]
            for m in ("1", "2") if paired else ("1",):
                rows = d["seq" + m][xt].clone()
                rows = torch.where(rows > 0, rows | 0x20, rows).contiguous()
                off = (torch.arange(len(xu), dtype=torch.int64, device=dev) * rows.shape[1]).to(torch.int32)
                xkeep += [rows, off]
        del d
        eng = engine.GpuEngine(params, device=dev.index or 0)
        r1 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
        r2 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
        pr = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
        nc = torch.zeros(1, dtype=torch.int32, device=dev)
        b = abi.Batch()
        b.n, b.flags = n, abi.BATCH_STAT_ISIZE
        b.seq1, b.qual1, b.len1 = (x.data_ptr() for x in bufs["1"])
        if paired:
            b.seq2, b.qual2, b.len2 = (x.data_ptr() for x in bufs["2"])
        if xkeep:
            b.n_exotic, b.exotic_unit = len(xkeep[0]), xkeep[0].ctypes.data
            for k in range(2 if paired else 1):
                b.exotic_text[k], b.exotic_off[k] = xkeep[1 + 2 * k].data_ptr(), xkeep[2 + 2 * k].data_ptr()
        r = abi.Results()
        r.r1 = r1.data_ptr()
        if paired:
            r.r2, r.pair = r2.data_ptr(), pr.data_ptr()
        r.n_corrections = nc.data_ptr()
        if params.n_adapter_fasta:   # --adapter_fasta: the trims' event list (the host replays FilterResult's adapter map from it)
            ev = torch.zeros(4 * n * 12, dtype=torch.uint8, device=dev)
            nev = torch.zeros(1, dtype=torch.int32, device=dev)
            r.adapter_events, r.adapter_events_capacity, r.n_adapter_events = ev.data_ptr(), 4 * n, nev.data_ptr()
        torch.cuda.synchronize()
        for _ in range(3):   # (warm-up: the first launches of an engine pay its lazy allocations and the clocks' ramp)
            eng.submit_device(b, r)
            eng.reset()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.submit_device(b, r)
            eng.reset()
        eng.synchronize()
        dt_all = (time.perf_counter() - t0) / steps
        # Every step re-submits the same batch, so the bloom filter and the counters are cleared after each (a run's per-RUN
        # work: 1 - 4 GiB of bitmaps) - timed apart and taken out of the per-step figure, as the headline amortises it over its run
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.reset()
        eng.synchronize()
        dt_reset = (time.perf_counter() - t0) / steps
        dt = max(dt_all - dt_reset, 1e-9)
        reads = n * (2 if paired else 1)
        # the HBM roofline of the whole step (kernels + folds + Duplicate's tail): algorithmic bytes of SURVEY.md 8(d) / wall / 8 TB/s
        bpp = algorithmic_bytes_per_pair(Lr) if paired else algorithmic_bytes_per_pair(Lr) // 2
        res.append({"config": name, "units_per_step": n, "ms_per_step": round(dt * 1e3, 3), "reset_ms_per_run": round(dt_reset * 1e3, 3),
                    "Mreads_per_s": round(reads / dt / 1e6, 1), "plan": eng.plan(),
                    "algorithmic_GBps": round(n * bpp / dt / 1e9, 1), "frac": round(n * bpp / dt / 1e9 / HBM_PEAK_GBPS, 4)})
        eng.close()
        del bufs, r1, r2, pr
        torch.cuda.empty_cache()

    p = abi.default_params(False, 150)
    p.adapter_seq_r1 = None
    p.adapter_enabled = 0
    p.poly_g = 1
    p.cut_right = 1
    run("configs[1]: SE 1x150, 10 M reads, -A -g --cut_right", p, 150, 10_000_000, False)
    # fastp's DEFAULT single-end run: the adapter is auto-detected and trimmed by sequence (seprocessor.cpp:240-252,
    # AdapterTrimmer::trimBySequence) - the TruSeq adapter the synthetic reads carry, as the Evaluator would report it
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    p = abi.default_params(False, 150)
    p.adapter_seq_r1 = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    run("SE 1x150 default (adapter auto-detected -> trimBySequence), 10 M reads", p, 150, 10_000_000, False)
    p = abi.default_params(True, 150)
    p.adapter_seq_r1 = b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA"
    p.adapter_seq_r2 = b"AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"
    p.cut_right = 1
    run("PE 2x150 --adapter_sequence/--adapter_sequence_r2 + --cut_right, 4 Mi pairs", p, 150, 4 * 1024 * 1024, True)
    # everyday flags that move or edit kept bases, all on the lane plan since round 5: -f / a UMI (the same front for every read
    # that is written out), -c, --merge (which switches -c on, options.cpp:119-121)
    p = abi.default_params(True, 150)
    p.cut_right = 1
    p.trim_front1 = p.trim_front2 = 5
    run("PE 2x150 -f 5 -F 5 --cut_right, 4 Mi pairs", p, 150, 4 * 1024 * 1024, True)
    p = abi.default_params(True, 150)
    p.cut_right = 1
    p.umi_len1, p.umi_len2 = 8, 0
    run("PE 2x150 --umi --umi_loc read1 --umi_len 8 --cut_right, 4 Mi pairs", p, 150, 4 * 1024 * 1024, True)
    p = abi.default_params(True, 150)
    p.cut_right = 1
    p.correction = 1
    run("PE 2x150 -c --cut_right, 4 Mi pairs", p, 150, 4 * 1024 * 1024, True)
    p = abi.default_params(True, 150)
    p.cut_right = 1
    p.merge = 1
    p.correction = 1
    run("PE 2x150 --merge --cut_right, 4 Mi pairs", p, 150, 4 * 1024 * 1024, True)
    # the option families that are not on the lane plan (or were not until round 6): what they cost
    p = abi.default_params(True, 150)
    p.cut_front = 1
    p.cut_tail = 1
    run("PE 2x150 --cut_front --cut_tail (-5 -3), 4 Mi pairs", p, 150, 4 * 1024 * 1024, True, steps=6)
    p = abi.default_params(True, 150)
    p.cut_right = 1
    p.allow_gap_overlap_trimming = 1
    run("PE 2x150 --allow_gap_overlap_trimming --cut_right, 4 Mi pairs", p, 150, 4 * 1024 * 1024, True, steps=6)
    p = abi.default_params(True, 150)
    p.cut_right = 1
    p.overlapped_out = 1
    run("PE 2x150 --overlapped_out --cut_right, 4 Mi pairs", p, 150, 4 * 1024 * 1024, True, steps=6)
    try:
        p = abi.default_params(True, 150)
        p.cut_right = 1
        abi.set_adapter_fasta(p, [b"AGATCGGAAGAGCACACGTCTGAACTCCAGTCA", b"AGATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT", b"CTGTCTCTTATACACATCT", b"TGGAATTCTCGGGTGCCAAGG"])
        run("PE 2x150 --adapter_fasta (4 adapters) --cut_right, 4 Mi pairs", p, 150, 4 * 1024 * 1024, True, steps=6)
    except Exception as e:   # noqa: BLE001
        res.append({"config": "PE 2x150 --adapter_fasta", "error": repr(e)[:200]})
    # letters outside ACGTN (soft-masked reads): one pair in a thousand goes through the text kernel, the rest through the lane plan
    p = abi.default_params(True, 150)
    p.cut_right = 1
    run("PE 2x150 --cut_right, 4 Mi pairs, 1 pair in 1000 soft-masked (lower case: the text kernel)", p, 150, 4 * 1024 * 1024, True,
        soft_masked_every=1000)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import cases
        import evalport
        Lr = 250
        p = abi.default_params(True, Lr)
        p.cut_right = 1
        p.dedup = 1
        d = synth_torch.synth_pairs_torch(20000, L=Lr, seed=5, device="cpu")
        pad = lambda a: np.pad(a.numpy(), ((0, 0), (0, 6)))
        b1 = cases._ArrayBatch(pad(d["seq1"]), d["len1"].numpy())
        b2 = cases._ArrayBatch(pad(d["seq2"]), d["len2"].numpy())
        e1, e2 = evalport.evaluate_seq_len(b1), evalport.evaluate_seq_len(b2)
        abi.set_overrep(p, evalport.evaluate_overrep_seqs(b1, e1), evalport.evaluate_overrep_seqs(b2, e2), e1, e2, 20)
        run(f"configs[4] per-GPU share: PE 2x250, 2 M pairs, --dedup -p ({p.n_overrep_seqs1}+{p.n_overrep_seqs2} seeds)", p, Lr, 2_000_000, True)
    except Exception as e:
        res.append({"config": "configs[4] per-GPU share", "error": repr(e)[:200]})
    return res


def log(msg):
    if os.environ.get("BENCH_VERBOSE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def respawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvp(cmd[0], cmd)


class ResidentBatch:
    """one batch of B pairs in HBM: packed rows + the abi.Batch that points at them"""

    def __init__(self, B, seed, dev):
        import torch
        import synth_torch
        from fastp_amd import abi
        d = synth_torch.synth_pairs_torch(B, L=L, seed=seed, device=dev)
        self.t = synth_torch.pack_torch(d["seq1"], d["qual1"], d["len1"], L) + synth_torch.pack_torch(d["seq2"], d["qual2"], d["len2"], L)
        del d
        b = abi.Batch()
        b.n, b.flags = B, abi.BATCH_STAT_ISIZE
        b.seq1, b.qual1, b.len1 = (x.data_ptr() for x in self.t[:3])
        b.seq2, b.qual2, b.len2 = (x.data_ptr() for x in self.t[3:])
        self.batch = b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=720, help="timed steps; a step = one batch through the hot path")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=0,
                    help="pairs per step per GPU; 0 = 4 Mi, or - fewer steps than batches - what makes ONE run of `steps` steps "
                         "hold 100.7 M pairs (BASELINE configs[2] at its full size)")
    ap.add_argument("--batches", type=int, default=24,
                    help="distinct batches resident per GPU = one run (24 x 4 Mi = 100.7 M pairs: BASELINE configs[2]); steps "
                         "cycle through them and the engine starts a new run (fresh bloom filter, counters) after each cycle")
    ap.add_argument("--no-extras", action="store_true", help="skip the end-to-end legs and the other configurations")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    # Every FASTP_GPU_* variable of the environment is an A/B switch of the library: a bench line says which ones it ran
    # with (`config.switches`), and does not run at all with the profiling build's ablation switch - its results are meaningless.
    switches = {k: v for k, v in sorted(os.environ.items()) if k.startswith("FASTP_GPU_") or k == "GPU_MAX_HW_QUEUES"}
    if int(os.environ.get("FASTP_GPU_DEBUG_SKIP", "0") or 0) != 0 and not os.environ.get("BENCH_ALLOW_ABLATION"):
        raise SystemExit("bench.py: FASTP_GPU_DEBUG_SKIP is set - a kernel with steps left out is not a bench line "
                         "(BENCH_ALLOW_ABLATION=1 for the profiling runs under profiles/, which never produce BENCH_*.json)")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        respawn_ranks(args.gpus)           # does not return
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")

    import torch
    import __graft_entry__ as graft
    from fastp_amd import abi, engine, multigpu
    import synth_torch

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = os.environ.get("BENCH_BACKEND", "nccl")   # nccl == RCCL on ROCm; "gloo" only for rehearsals
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if backend != "nccl":
        local = local % torch.cuda.device_count()            # rehearsal of the N>1 path on fewer GPUs
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if rank == 0:
        graft.build()
    if dist is not None:
        dist.barrier()

    params, ref_flags = bench_params()
    log("creating engine")
    eng = engine.GpuEngine(params, device=local)
    NB = max(1, min(args.batches, max(args.steps, args.warmup)))
    FULL = 24 * 4 * 1024 * 1024      # BASELINE configs[2]: 100 M pairs (here 100,663,296) in one run
    if args.pairs > 0:
        B = args.pairs
    elif NB < args.batches:          # a short run (the driver's --steps 20): fewer, larger batches - still 100.7 M pairs per run
        B = -(-FULL // NB // 64) * 64
    else:
        B = 4 * 1024 * 1024
    log(f"generating {NB} resident batches of {B} pairs")
    t_gen = time.perf_counter()
    resident = [ResidentBatch(B, 42 + 1000 * rank + k, dev) for k in range(NB)]
    torch.cuda.synchronize(dev)
    t_gen = time.perf_counter() - t_gen      # outside the timed region; every rank generates its own shard on its own GPU
    r1 = torch.zeros(B * 12, dtype=torch.uint8, device=dev)
    r2 = torch.zeros(B * 12, dtype=torch.uint8, device=dev)
    pr = torch.zeros(B * 8, dtype=torch.uint8, device=dev)
    ncorr = torch.zeros(1, dtype=torch.int32, device=dev)
    res = abi.Results()
    res.r1, res.r2, res.pair = r1.data_ptr(), r2.data_ptr(), pr.data_ptr()
    res.corrections, res.corrections_capacity, res.n_corrections = None, 0, ncorr.data_ptr()
    torch.cuda.synchronize(dev)
    log("warmup")

    # N>1: the exact sharded protocol - duplicate scan pass, bitmap prefix exchange, decision pass -
    # so that the merged result is that of one stream.  BENCH_SHARD=plain runs the per-shard submit instead
    # (cross-shard duplicates missed); BENCH_SHARD=force runs the protocol at N=1 too.
    shard_mode = os.environ.get("BENCH_SHARD", "exact")
    protocol = shard_mode == "force" or (shard_mode == "exact" and world > 1)

    # per-step duplicate scan state (17 B/pair), allocated once outside the timed region
    scan_bufs = [torch.empty(max(16, eng.dup_scan_bytes(B)), dtype=torch.uint8, device=dev) for _ in range(NB)] if protocol else []

    def run_steps(k):
        """k steps = k batches, cycling through the resident set; a cycle is one run of the engine"""
        done = 0
        while done < k:
            m = min(NB, k - done)
            if protocol:
                multigpu.run_shard(eng, dist, rank, world, [rb.batch for rb in resident[:m]], [res] * m, dev, force=True,
                                   scans=scan_bufs[:m], exchange=exchange["how"], timings=timings)
            else:
                for rb in resident[:m]:
                    eng.submit_device(rb.batch, res)
            done += m
            if done < k:
                eng.reset()     # next run: fresh Duplicate bitmaps / Stats / FilterResult (inside the timed region)

    def agree(failed):
        """all ranks take the same branch after a failure on any of them"""
        if dist is None:
            return failed
        t = torch.tensor([1 if failed else 0], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    # The collectives of an N > 1 run - the bitmap prefix exchange between the two passes and the counter all-reduce at the
    # end (Stats::merge / FilterResult::merge) - are the C ABI's own by default (fastp_gpu_comm_init + fastp_gpu_exchange_dup_prefix
    # + fastp_gpu_allreduce: RCCL behind the drop-in boundary, what a patched fastp would call).  They are set up and rehearsed
    # OUTSIDE the timed region under a watchdog: a native collective that hangs cannot be interrupted, so the rehearsal runs on
    # a helper thread and every rank votes; any failure or timeout on any rank -> all ranks use torch.distributed's
    # collectives (RCCL as well) instead.  BENCH_ALLREDUCE=torch / BENCH_EXCHANGE=torch skip the attempt.
    merge_how, exchange_how = "n/a", "n/a"
    use_cabi = False
    cerr = None
    if dist is not None:
        use_cabi = backend == "nccl" and os.environ.get("BENCH_ALLREDUCE", "cabi") == "cabi"
        if use_cabi:
            import threading
            box = {}

            def rehearse():
                try:
                    torch.cuda.set_device(dev)   # the current device is per thread
                    ids = [eng.comm_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    eng.comm_init(ids[0], world, rank)
                    eng.allreduce()
                    box["ok"] = True
                except Exception as e:   # noqa: BLE001
                    box["err"] = e
            th = threading.Thread(target=rehearse, daemon=True)
            th.start()
            th.join(float(os.environ.get("BENCH_CABI_TIMEOUT", "90")))
            if th.is_alive():
                cerr = TimeoutError("fastp_gpu_allreduce rehearsal did not return")
            elif "err" in box:
                cerr = box["err"]
            if agree(cerr is not None):
                print(f"[bench] fastp_gpu_allreduce unavailable on rank {rank} ({cerr!r}); using torch.distributed", file=sys.stderr, flush=True)
                use_cabi = False
        merge_how = "fastp_gpu_allreduce (RCCL, C ABI)" if use_cabi else \
                    f"torch.distributed all_reduce ({backend})" + (f"; C ABI attempt failed: {type(cerr).__name__}" if cerr else "")
    exchange = {"how": "cabi" if (use_cabi and os.environ.get("BENCH_EXCHANGE", "cabi") == "cabi") else "torch"}
    timings = {}

    err = None
    try:
        if protocol and exchange["how"] == "cabi":   # the first exchange under the watchdog too
            import threading
            box = {}

            def first():
                try:
                    torch.cuda.set_device(dev)
                    run_steps(args.warmup)
                    box["ok"] = True
                except Exception as e:   # noqa: BLE001
                    box["err"] = e
            th = threading.Thread(target=first, daemon=True)
            th.start()
            th.join(float(os.environ.get("BENCH_CABI_TIMEOUT", "90")) + 60.0)
            xerr = TimeoutError("fastp_gpu_exchange_dup_prefix did not return") if th.is_alive() else box.get("err")
            if agree(xerr is not None):
                print(f"[bench] fastp_gpu_exchange_dup_prefix unavailable on rank {rank} ({xerr!r}); exchange through torch.distributed", file=sys.stderr, flush=True)
                if th.is_alive():
                    raise SystemExit("bench.py: a native collective hangs; rerun with BENCH_EXCHANGE=torch")
                exchange["how"] = "torch"
                eng.reset()
                run_steps(args.warmup)
        else:
            run_steps(args.warmup)
    except Exception as e:   # e.g. a collective the installed RCCL build refuses: keep measuring, say so in the JSON line
        if not protocol:
            raise
        err = e
    if protocol and agree(err is not None):
        print(f"[bench] exact sharded protocol failed on rank {rank} ({err!r}); falling back to per-shard submit", file=sys.stderr, flush=True)
        protocol = False
        shard_mode = "plain (exact protocol failed: %s)" % (type(err).__name__ if err else "on another rank")
        eng.reset()
        run_steps(args.warmup)
    if dist is not None and protocol:
        exchange_how = "fastp_gpu_exchange_dup_prefix (RCCL, C ABI)" if exchange["how"] == "cabi" else f"torch.distributed all_to_all_single ({backend})"
    if dist is not None and not use_cabi:
        multigpu.allreduce_counters_device(eng, dist, dev)

    def merge_counters():
        if dist is None:
            return
        if use_cabi:
            eng.allreduce()
        else:
            multigpu.allreduce_counters_device(eng, dist, dev)

    eng.synchronize()
    eng.reset()
    eng.kernel_time()  # reset the event accumulator
    timings.clear()

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run_steps(args.steps)
    eng.synchronize()
    t_merge = time.perf_counter()
    merge_counters()                                      # Stats::merge / FilterResult::merge
    eng.synchronize()
    timings["merge_s"] = time.perf_counter() - t_merge
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kms, klaunches = eng.kernel_time()
    plan = eng.plan()
    log(f"timed region done: {elapsed:.3f}s, kernel {kms:.2f} ms over {klaunches} launches")
    if rank == 0:
        total_pairs = B * args.steps * world
        value = 2.0 * total_pairs / elapsed / 1e6
        per_launch_pairs = B * args.steps / max(1, klaunches)
        avg_ms = kms / max(1, klaunches)
        bpp = algorithmic_bytes_per_pair(L)
        achieved = per_launch_pairs * bpp / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM-side bytes and VALU instructions come from the committed rocprofv3 --pmc passes (profiles/traffic.json,
        # profiles/valu.json: figures PER PAIR of the same kernels, so they apply at any batch size), durations are live
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                if "hbm_bytes_per_pair" in tj:
                    traffic = round(tj["hbm_bytes_per_pair"] * per_launch_pairs)
                elif tj.get("pairs_per_launch"):
                    traffic = round(tj["hbm_bytes_per_launch"] / tj["pairs_per_launch"] * per_launch_pairs)
            except Exception:
                traffic = None
        # what actually bounds the kernels: integer VALU issue, reported next to the HBM roofline BASELINE.json asks for
        compute = None
        vf = os.path.join(ROOT, "profiles", "valu.json")
        if os.path.exists(vf) and avg_ms > 0:
            try:
                vj = json.load(open(vf))
                ipp = vj.get("insts_valu_per_pair") or vj["insts_valu_per_launch"] / vj["pairs_per_launch"]
                ach = ipp * per_launch_pairs * 64 / (avg_ms * 1e-3) / 1e12
                compute = {"bound": "valu_int", "achieved": round(ach, 2), "peak": vj["peak_T_lane_ops_per_s"],
                           "unit": "T lane-ops/s", "frac": round(ach / vj["peak_T_lane_ops_per_s"], 4),
                           "insts_valu_per_pair": round(ipp, 1), "profile": vj.get("tag")}
            except Exception:
                compute = None
        runs = args.steps / NB
        out = {
            "metric": "Mreads/sec (whole node), 2x150 bp PE, inputs resident in HBM", "value": round(value, 3),
            "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: PE 2x150 bp synthetic (fragment model), {NB * B / 1e6:.1f} M distinct pairs per GPU "
                                   f"per run ({NB} resident batches of {B}), auto-adapter via overlap + --cut_right quality trim, dup "
                                   f"evaluation on, fastp default filters; {runs:.2f} runs timed back to back (fresh bloom filter and "
                                   f"counters per run)",
                       "pairs_per_run_per_gpu": NB * B, "pairs_per_step_per_gpu": B, "timed_pairs_total": total_pairs,
                       "read_len": L, "parallelism": f"shard x{world}", "counter_merge": merge_how,
                       "generate_s": round(t_gen, 2),
                       # per-run constants of an N > 1 run, inside the timed region but reported apart so that a 1 -> N curve
                       # can be read: the bitmap prefix exchange (once per run of NB steps) and the counter all-reduce (once)
                       "bitmap_exchange": exchange_how,
                       "exchange_ms_per_run": round(timings.get("exchange_s", 0.0) / max(1.0, -(-args.steps // NB)) * 1e3, 3) if protocol else None,
                       "merge_ms": round(timings.get("merge_s", 0.0) * 1e3, 3) if dist is not None else None,
                       "cross_shard_duplicates": "exact (scan pass + bitmap prefix exchange + decision pass)" if protocol else
                                                 ("n/a" if world == 1 else "per shard: " + shard_mode)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "kernel": KERNELS.get(plan, "fq_fused_kernel"), "kernel_avg_ms": round(avg_ms, 4),
                         "algorithmic_bytes_per_pair": bpp, "pairs_per_launch": int(per_launch_pairs)},
        }
        if compute is not None:
            out["compute_roofline"] = compute
        out["config"]["kernel_plan"] = plan
        out["config"]["switches"] = switches   # {} = the library's defaults
        if world == 1:
            # the legs below make engines of their own: this one (its streams, bloom filter and the resident batches) is done
            eng.close()
            eng = None
        if world == 1 and not args.no_cpu:
            resident = None
            torch.cuda.empty_cache()
            have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "fastp_ref"))
            files = write_sample_files(args.cpu_sample, dev) if have_ref else None
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, ref_flags, params, dev, files)
            if files is not None and not args.no_extras:
                out.update(e2e_legs(args.cpu_sample, ref_flags, params, files, out["cpu_baseline"]["value"], full_json=True))
            if files is not None:
                import shutil
                shutil.rmtree(files[0], ignore_errors=True)
            # the same leg on a sample three times the size: a run's start-up (process, HIP runtime, engine, page-locked buffers:
            # startup_s above) is most of a 4 M-pair run's wall clock, so the larger sample shows the rate a real file sees
            if files is not None and not args.no_extras and "e2e_dropin" in out and out["e2e_dropin"].get("gpu"):
                big = None
                try:
                    big = write_sample_files(3 * args.cpu_sample, dev)
                    ref = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")
                    t0 = time.time()
                    subprocess.run([ref, "-i", big[1], "-I", big[2], "-o", os.path.join(big[0], "o1.fq"), "-O", os.path.join(big[0], "o2.fq"), "-j",
                                    os.path.join(big[0], "o.json"), "-h", os.path.join(big[0], "o.html"), "-w", str(host_cores())] + ref_flags,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=900)
                    cpu_big = round(2 * 3 * args.cpu_sample / (time.time() - t0) / 1e6, 4)
                    leg = e2e_legs(3 * args.cpu_sample, ref_flags, params, big, cpu_big, dropin_only=True)
                    out["e2e_dropin_large"] = dict(leg.get("e2e_dropin", {}), pairs=3 * args.cpu_sample)
                except Exception as e:   # never at the expense of the kernel line
                    out["e2e_dropin_large"] = {"gpu": None, "error": repr(e)[:200]}
                finally:
                    if big is not None:
                        import shutil
                        shutil.rmtree(big[0], ignore_errors=True)
        if world == 1 and not args.no_extras:
            resident = None
            torch.cuda.empty_cache()
            try:
                out["other_configs"] = other_configs(dev)
            except Exception as e:
                out["other_configs"] = [{"error": repr(e)[:200]}]
            if not args.no_cpu:   # last: nothing above depends on it
                try:
                    leg = e2e_compressed_leg(args.cpu_sample, ref_flags, dev)
                    if leg is not None:
                        out["e2e_dropin_bgzf"] = leg
                except Exception as e:
                    out["e2e_dropin_bgzf"] = {"gpu": None, "error": repr(e)[:300]}
        print(json.dumps(out))
    if eng is not None:
        eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
