"""Throughput of the device FASTQ parser (fastp_gpu_parse_fastq) on synthetic 150 bp records."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import numpy as np, torch
from fastp_amd import abi, engine
import synth
dev = torch.device('cuda', 0)
n0 = 20000
d = synth.synth_pairs(n0, L=150, seed=1, paired=False, ragged_frac=0.0)
txt = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
text = txt * reps
n = n0 * reps
pad = (-len(text)) % 16 + 16
t = torch.frombuffer(bytearray(text + b"\0" * pad), dtype=torch.uint8).to(dev)
g = engine.GpuEngine(abi.default_params(False, 150))
ss, qs = abi.seq_stride(150), abi.qual_stride(150)
seq = torch.empty((n, ss), dtype=torch.uint8, device=dev); qual = torch.empty((n, qs), dtype=torch.uint8, device=dev)
lens = torch.empty(n, dtype=torch.int16, device=dev)
loff = torch.empty(4 * n, dtype=torch.int32, device=dev); llen = torch.empty(4 * n, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    info = g.parse_fastq(t.data_ptr(), len(text), True, n, seq.data_ptr(), qual.data_ptr(), lens.data_ptr(), loff.data_ptr(), llen.data_ptr())
    dt = time.perf_counter() - t0
    print(f"{info.n_records} records, {len(text)/1e6:.1f} MB in {dt*1e3:.2f} ms -> {info.n_records/dt/1e6:.1f} Mreads/s, {len(text)/dt/1e9:.2f} GB/s")
