"""The drop-in end to end: the real reference with its worker loops bound to the engine (oracle/_ref/fastp_ref_gpu,
FASTP_GPU=1) next to the unpatched reference on the same plain FASTQ files (tmpfs), several thread counts and window
sizes; output md5s and the JSON reports compared (against `fastp_ref -w 1`).
usage: python tools/dropin_bench.py [--pairs N]"""
import argparse, hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tools")
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=4_000_000)
args = ap.parse_args()
dev = torch.device("cuda", 0)
tmp, f1, f2 = bench.write_sample_files(args.pairs, dev)
REF = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")
GPU = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu")
flags = ["-G", "--cut_right"]


def md5(p):
    h = hashlib.md5()
    with open(p, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def run(binary, w, tag, env=None):
    cmd = [binary, "-i", f1, "-I", f2, "-o", f"{tmp}/{tag}1.fq", "-O", f"{tmp}/{tag}2.fq", "-j", f"{tmp}/{tag}.json", "-h", f"{tmp}/{tag}.html",
           "-w", str(w)] + flags
    best = None
    for _ in range(2):
        t0 = time.time()
        p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})), timeout=1200)
        if p.returncode != 0:
            return None, p.stderr.decode()[-400:]
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
    rep = json.load(open(f"{tmp}/{tag}.json"))
    rep.pop("command", None)
    return best, (md5(f"{tmp}/{tag}1.fq"), md5(f"{tmp}/{tag}2.fq"), rep)


print(f"{args.pairs} synthetic 2x150 pairs, plain FASTQ on {os.path.dirname(tmp)}; host has {os.cpu_count()} logical cores", flush=True)
t1, r1 = run(REF, 1, "ref1")
print(f"fastp_ref     -w  1         : {t1:.2f} s = {2*args.pairs/t1/1e6:.2f} Mreads/s", flush=True)
base = r1
for w in (2, 4, 8, 16):
    t, r = run(REF, w, f"ref{w}")
    same = r[0] == base[0] and r[1] == base[1]
    print(f"fastp_ref     -w {w:2d}         : {t:.2f} s = {2*args.pairs/t/1e6:.2f} Mreads/s  outputs == -w 1: {same}", flush=True)
for w, packs in ((1, 32), (2, 32), (4, 32), (8, 32), (16, 32), (4, 16), (4, 64), (16, 16), (16, 64)):
    t, r = run(GPU, w, f"gpu{w}_{packs}", {"FASTP_GPU": "1", "FASTP_GPU_PACKS": str(packs)})
    if t is None:
        print(f"fastp_ref_gpu -w {w:2d} packs {packs}: FAILED {r}", flush=True)
        continue
    same = r[0] == base[0] and r[1] == base[1]
    diff = [k for k in base[2] if base[2][k] != r[2].get(k)]
    print(f"fastp_ref_gpu -w {w:2d} packs {packs:3d}: {t:.2f} s = {2*args.pairs/t/1e6:.2f} Mreads/s  outputs == fastp_ref -w 1: {same}; "
          f"JSON sections that differ from fastp_ref -w 1: {diff or 'none'}", flush=True)
# where the time goes in a GPU run: the engine's start-up (HIP runtime, 1 GiB of bloom bitmaps) and the reference's own reader
t0 = time.time()
subprocess.run([GPU, "-i", f1, "-I", f2, "-o", f"{tmp}/x1.fq", "-O", f"{tmp}/x2.fq", "-j", f"{tmp}/x.json", "-h", f"{tmp}/x.html", "-w", "4",
                "--reads_to_process", "1000"] + flags, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, FASTP_GPU="1"))
print(f"fastp_ref_gpu on the first 1000 pairs only (start-up + tear-down): {time.time()-t0:.2f} s", flush=True)
t0 = time.time()
subprocess.run([REF, "-i", f1, "-I", f2, "-o", f"{tmp}/x1.fq", "-O", f"{tmp}/x2.fq", "-j", f"{tmp}/x.json", "-h", f"{tmp}/x.html", "-w", "4",
                "--reads_to_process", "1000"] + flags, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
print(f"fastp_ref     on the first 1000 pairs only (start-up + tear-down): {time.time()-t0:.2f} s", flush=True)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
