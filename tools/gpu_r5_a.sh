#!/bin/bash
# round 5, first GPU visit: what the last session of round 4 could only run on the emulator (".gz" inputs of the stream binding),
# then where a bgzip-compressed run's time goes as a function of the trip size, then the host inflater of plain gzip inputs
# (fq_pgunzip.h) on the GPU box's own cores: alone (tools/gunzip_bench.cpp) and inside the drop-in, by thread count.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_a.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_zz_gpu_compressed_inputs.py -m gpu -q --durations=25 > gpurun_out/r5a_pytest_gz.log 2>&1; echo "pytest (compressed inputs) rc=$?"; tail -3 gpurun_out/r5a_pytest_gz.log
# the drop-in on its own .gz output, 12 M pairs, trip sizes 16 .. 128 MiB (members per inflate launch: ~520 .. ~4100)
timeout 800 python - > gpurun_out/r5a_bgzf_chunk_sweep.txt 2>&1 <<'PY'
import os, re, subprocess, sys, time, shutil
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
dev = torch.device("cuda:0")
params, flags = bench.bench_params()
pairs = 12_000_000
tmp, f1, f2 = bench.write_sample_files(pairs, dev)
J = lambda n: os.path.join(tmp, n)
gpu = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu")
ref = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")
def run(binary, i1, i2, tag, ext, env):
    cmd = [binary, "-i", i1, "-I", i2, "-o", J(tag + "1" + ext), "-O", J(tag + "2" + ext), "-j", J(tag + ".json"), "-h", J(tag + ".html"), "-w", "16"] + flags
    t0 = time.time()
    pr = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, **env))
    return time.time() - t0, pr.returncode, pr.stderr.decode(errors="replace")
g = {"FASTP_GPU": "1", "FASTP_GPU_VERBOSE": "1"}
t, rc, err = run(gpu, f1, f2, "z", ".fq.gz", g)
print(f"plain in -> .gz out: {t:.2f} s rc={rc}"); print("\n".join(l for l in err.splitlines() if "stream mode" in l))
os.remove(f1); os.remove(f2)
for mb in (16, 32, 64, 128):
    for inflate in ("", "lane"):
        env = dict(g, FASTP_GPU_STREAM_CHUNK_MB=str(mb))
        if inflate:
            env["FASTP_GPU_INFLATE"] = inflate
        t, rc, err = run(gpu, J("z1.fq.gz"), J("z2.fq.gz"), "b", ".fq", env)
        print(f"chunk {mb} MiB inflate={inflate or 'auto'}: {t:.2f} s rc={rc}"); print("\n".join(l for l in err.splitlines() if "stream mode" in l), flush=True)
t, rc, err = run(ref, J("z1.fq.gz"), J("z2.fq.gz"), "c", ".fq", {})
print(f"fastp_ref -w 16 on the same .gz files (zlib behind the ISA-L shim): {t:.2f} s rc={rc}")
import hashlib
md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
print("outputs identical:", md5(J("b1.fq")) == md5(J("c1.fq")) and md5(J("b2.fq")) == md5(J("c2.fq")))
shutil.rmtree(tmp, ignore_errors=True)
PY
echo "sweep rc=$?"; tail -30 gpurun_out/r5a_bgzf_chunk_sweep.txt
# plain gzip inputs (what sequencers deliver): several host threads per file; 4 M pairs, gzip -1 of the bench's sample files
timeout 700 python - > gpurun_out/r5a_plain_gzip_inputs.txt 2>&1 <<'PY'
import os, subprocess, sys, time, shutil, hashlib
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
dev = torch.device("cuda:0")
params, flags = bench.bench_params()
tmp, f1, f2 = bench.write_sample_files(4_000_000, dev)
J = lambda n: os.path.join(tmp, n)
t0 = time.time()
ps = [subprocess.Popen(["gzip", "-1", "-k", f]) for f in (f1, f2)]
[p.wait() for p in ps]
print(f"gzip -1 of both files: {time.time() - t0:.1f} s; {os.path.getsize(f1 + '.gz') / 1e6:.0f} MB each, nproc {os.cpu_count()}")
subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "gunzip_bench.cpp"), "-lz", "-o", J("gunzip_bench")], check=True)
out = subprocess.run([J("gunzip_bench"), f1 + ".gz", "1", "2", "4", "8", "12", "16", "24", "32"], capture_output=True, text=True).stdout.splitlines()
print("\n".join(out[:18]))
gpu = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu")
ref = os.path.join(ROOT, "oracle", "_ref", "fastp_ref")
def run(binary, i1, i2, tag, env):
    cmd = [binary, "-i", i1, "-I", i2, "-o", J(tag + "1.fq"), "-O", J(tag + "2.fq"), "-j", J(tag + ".json"), "-h", J(tag + ".html"), "-w", "16"] + flags
    t0 = time.time()
    pr = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, **env))
    return time.time() - t0, pr.returncode, pr.stderr.decode(errors="replace")
g = {"FASTP_GPU": "1", "FASTP_GPU_VERBOSE": "1"}
t, rc, err = run(gpu, f1, f2, "p", g)
print(f"plain text in: {t:.2f} s rc={rc}")
for th in ("1", "2", "4", "8", "12", "16"):
    t, rc, err = run(gpu, f1 + ".gz", f2 + ".gz", "g", dict(g, FASTP_GPU_STREAM_GUNZIP_THREADS=th))
    print(f"gzip in, {th} inflater thread(s) per file: {t:.2f} s = {8.0 / t:.1f} Mreads/s rc={rc}"); print("\n".join(l for l in err.splitlines() if "stream mode" in l), flush=True)
t, rc, err = run(ref, f1 + ".gz", f2 + ".gz", "c", {})
print(f"fastp_ref -w 16 on the same .gz files (zlib behind the ISA-L shim): {t:.2f} s = {8.0 / t:.1f} Mreads/s rc={rc}")
md5 = lambda p: hashlib.md5(open(p, "rb").read()).hexdigest()
print("outputs identical:", md5(J("g1.fq")) == md5(J("c1.fq")) and md5(J("g2.fq")) == md5(J("c2.fq")))
shutil.rmtree(tmp, ignore_errors=True)
PY
echo "plain gzip rc=$?"; tail -30 gpurun_out/r5a_plain_gzip_inputs.txt
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5a_bench_driver_cmd.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/r5a_bench_driver_cmd.log | cut -c1-7000
