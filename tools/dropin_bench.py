"""The drop-in end to end: the real reference bound to the engine (oracle/_ref/fastp_ref_gpu, FASTP_GPU=1) next to the
unpatched reference on the same plain FASTQ files (tmpfs).  Stream mode (raw chunks -> device parser -> worker loop ->
device formatter -> the writers' files; the default), its WriterThread::input hand-off, pack mode (the reference's own
reader threads + the worker-loop hook), several thread counts; output md5s and the whole JSON report compared with
`fastp_ref -w 1`; the reference at -w N against itself at -w 1 (which JSON sections depend on the thread count).
usage: python tools/dropin_bench.py [--pairs N] [--big N2]"""
import argparse, hashlib, json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tools")
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=4_000_000)
ap.add_argument("--big", type=int, default=0, help="a second, larger sample for the stream mode (start-up amortised)")
ap.add_argument("--ref-threads", default="1,16")
ap.add_argument("--quick", action="store_true", help="stream mode only (default settings, the string hand-off, more write pieces)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
REF = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")
GPU = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu")
flags = ["-G", "--cut_right"]


def md5(p):
    h = hashlib.md5()
    with open(p, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def run(binary, w, tag, tmp, f1, f2, env=None, reps=2, extra=()):
    cmd = [binary, "-i", f1, "-I", f2, "-o", f"{tmp}/{tag}1.fq", "-O", f"{tmp}/{tag}2.fq", "-j", f"{tmp}/{tag}.json", "-h", f"{tmp}/{tag}.html",
           "-w", str(w)] + flags + list(extra)
    best, err = None, ""
    for _ in range(reps):
        t0 = time.time()
        try:
            p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})), timeout=300)
        except subprocess.TimeoutExpired:
            return None, "timeout (300 s)", ""
        if p.returncode != 0:
            return None, p.stderr.decode()[-400:], ""
        dt = time.time() - t0
        if best is None or dt < best:
            best, err = dt, p.stderr.decode(errors="replace")
    rep = json.load(open(f"{tmp}/{tag}.json"))
    rep.pop("command", None)
    out = (md5(f"{tmp}/{tag}1.fq"), md5(f"{tmp}/{tag}2.fq"), rep)
    for x in ("1.fq", "2.fq"):
        os.unlink(f"{tmp}/{tag}{x}")
    return best, out, err


def sections(a, b):
    return [k for k in a if a[k] != b.get(k)]


def sample(pairs):
    tmp, f1, f2 = bench.write_sample_files(pairs, dev)
    print(f"\n== {pairs} synthetic 2x150 pairs, plain FASTQ on {os.path.dirname(tmp)} ({(os.path.getsize(f1) + os.path.getsize(f2)) / 1e9:.2f} GB); "
          f"host has {os.cpu_count()} logical cores", flush=True)
    return tmp, f1, f2


def stream_line(err):
    m = re.search(r"fastp_gpu: stream mode: .*", err)
    return m.group(0) if m else "(no stream line)"


def block(pairs, ref_threads, gpu_cfgs):
    tmp, f1, f2 = sample(pairs)
    base = None
    for w in ref_threads:
        t, r, _ = run(REF, w, f"ref{w}", tmp, f1, f2, reps=1 if w == 1 else 2)
        if base is None:
            base = r
        same = r[0] == base[0] and r[1] == base[1]
        print(f"fastp_ref     -w {w:2d}              : {t:7.2f} s = {2*pairs/t/1e6:6.2f} Mreads/s  outputs == -w 1: {same}; JSON sections != -w 1: {sections(base[2], r[2]) or 'none'}", flush=True)
    for name, w, env in gpu_cfgs:
        e = dict({"FASTP_GPU": "1", "FASTP_GPU_VERBOSE": "1"}, **env)
        t, r, err = run(GPU, w, "gpu", tmp, f1, f2, env=e)
        if t is None:
            print(f"fastp_ref_gpu -w {w:2d} {name:12s}: FAILED {r}", flush=True)
            continue
        same = r[0] == base[0] and r[1] == base[1]
        print(f"fastp_ref_gpu -w {w:2d} {name:12s}: {t:7.2f} s = {2*pairs/t/1e6:6.2f} Mreads/s  outputs == fastp_ref -w 1: {same}; JSON sections != fastp_ref -w 1: "
              f"{sections(base[2], r[2]) or 'none'}", flush=True)
        if "stream" in name:
            print("      " + stream_line(err), flush=True)
    # start-up + tear-down of each binary: the first 1000 pairs only
    for binary, env, label in ((GPU, {"FASTP_GPU": "1"}, "fastp_ref_gpu (stream)"), (REF, {}, "fastp_ref")):
        t0 = time.time()
        subprocess.run([binary, "-i", f1, "-I", f2, "-o", f"{tmp}/x1.fq", "-O", f"{tmp}/x2.fq", "-j", f"{tmp}/x.json", "-h", f"{tmp}/x.html", "-w", "4",
                        "--reads_to_process", "1000"] + flags, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
        print(f"{label} on the first 1000 pairs only (start-up + tear-down): {time.time()-t0:.2f} s", flush=True)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


cfgs = [("stream", 16, {}), ("stream", 4, {}),
        ("stream c64", 16, {"FASTP_GPU_STREAM_CHUNK_MB": "64"}), ("stream c16", 16, {"FASTP_GPU_STREAM_CHUNK_MB": "16"}),
        ("stream io16", 16, {"FASTP_GPU_STREAM_IO_THREADS": "16"}),
        ("stream input", 16, {"FASTP_GPU_WRITER": "input"}),
        ("pack", 16, {"FASTP_GPU_STREAM": "0"})]
if args.quick:
    cfgs = [("stream", 16, {}), ("stream c32", 16, {"FASTP_GPU_STREAM_CHUNK_MB": "32"}), ("stream wp8", 16, {"FASTP_GPU_STREAM_WRITE_PIECE_MB": "8"}),
            ("stream input", 16, {"FASTP_GPU_WRITER": "input"})]
block(args.pairs, [int(x) for x in args.ref_threads.split(",")], cfgs)
if args.big:
    block(args.big, [1, 16], [("stream", 16, {}), ("stream c32", 16, {"FASTP_GPU_STREAM_CHUNK_MB": "32"}), ("stream wp8", 16, {"FASTP_GPU_STREAM_WRITE_PIECE_MB": "8"}),
                              ("stream input", 16, {"FASTP_GPU_WRITER": "input"})])
