"""End to end on files: plain FASTQ in -> FASTQ out through fastp_amd.pipeline (parse, worker loop, format all on the
GPU), timed next to reference fastp (oracle/_ref/fastp_ref) on the same files, outputs compared by md5.
usage: python tools/e2e_fastq.py [pairs] [chunk_MiB]"""
import hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, torch
from fastp_amd import abi, pipeline
import synth_torch

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
chunk = (int(sys.argv[2]) if len(sys.argv) > 2 else 256) << 20
L = 150
base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
tmp = tempfile.mkdtemp(prefix="fastp_e2e_", dir=base)
f1, f2 = tmp + "/r1.fq", tmp + "/r2.fq"
t0 = time.time()
with open(f1, "wb") as a, open(f2, "wb") as b:
    # one synthetic block of <= 1 M pairs, written as often as needed (repeats only add duplicates)
    k = min(pairs, 1_000_000)
    d = synth_torch.synth_pairs_torch(k, L=L, seed=7, device="cpu")
    t1, t2 = synth_torch.to_fastq_bytes(d["seq1"], d["qual1"], 1), synth_torch.to_fastq_bytes(d["seq2"], d["qual2"], 2)
    reps = (pairs + k - 1) // k
    pairs = reps * k
    for _ in range(reps):
        a.write(t1); b.write(t2)
print(f"input: {pairs} pairs, {os.path.getsize(f1) + os.path.getsize(f2)} bytes, generated in {time.time()-t0:.1f}s", flush=True)

p = abi.default_params(True, L); p.cut_right = 1
pl = pipeline.FastqPipeline(p, chunk_bytes=chunk)
def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()
res = {}
for rep in range(2):
    st = pl.run(f1, f2, tmp + "/g1.fq", tmp + "/g2.fq") if rep == 0 else None
    if st: break
res["gpu"] = st
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()}), flush=True)
print(f"GPU pipeline: {2*st['units']/st['wall']/1e6:.2f} Mreads/s end to end (wall {st['wall']:.2f}s)", flush=True)
ref = ROOT + "/oracle/_ref/fastp_ref"
if os.path.exists(ref):
    cores = min(os.cpu_count() or 1, 16)
    sample = pairs
    cmd = [ref, "-i", f1, "-I", f2, "-o", tmp + "/o1.fq", "-O", tmp + "/o2.fq", "-j", tmp + "/r.json", "-h", tmp + "/r.html",
           "-w", str(cores), "-G", "--cut_right"]
    t0 = time.time(); subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=3000); w = time.time() - t0
    print(f"reference fastp -w {cores}: {2*sample/w/1e6:.2f} Mreads/s (wall {w:.2f}s)", flush=True)
    same = md5(tmp + "/o1.fq") == md5(tmp + "/g1.fq") and md5(tmp + "/o2.fq") == md5(tmp + "/g2.fq")
    print("md5(out1), md5(out2) identical to the reference:", same, flush=True)
    print(f"speedup end to end: {w / st['wall']:.1f}x")
for fn in os.listdir(tmp):
    os.unlink(tmp + "/" + fn)
os.rmdir(tmp)
