"""Throughput of the device FASTQ formatter (fastp_gpu_format_fastq) on synthetic 2x150 bp records:
parse -> submit_device -> format, every buffer resident in HBM."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import numpy as np, torch
from fastp_amd import abi, engine
import synth
dev = torch.device('cuda', 0)
n0 = 20000
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
d = synth.synth_pairs(n0, L=150, seed=1, paired=True)
p = abi.default_params(True, 150)
p.correction = 1
g = engine.GpuEngine(p)
n = n0 * reps
ss, qs = abi.seq_stride(150), abi.qual_stride(150)
mates = []
for m in (1, 2):
    text = synth.to_fastq(d[f"seq{m}"], d[f"qual{m}"], d[f"len{m}"], m) * reps
    pad = (-len(text)) % 16 + 16
    t = torch.frombuffer(bytearray(text + b"\0" * pad), dtype=torch.uint8).to(dev)
    seq = torch.empty((n, ss), dtype=torch.uint8, device=dev); qual = torch.empty((n, qs), dtype=torch.uint8, device=dev)
    lens = torch.empty(n, dtype=torch.int16, device=dev)
    loff = torch.empty(4 * n, dtype=torch.int32, device=dev); llen = torch.empty(4 * n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    info = g.parse_fastq(t.data_ptr(), len(text), True, n, seq.data_ptr(), qual.data_ptr(), lens.data_ptr(), loff.data_ptr(), llen.data_ptr())
    assert info.n_records == n
    mates.append(dict(t=t, seq=seq, qual=qual, lens=lens, loff=loff, llen=llen, nbytes=len(text)))
res = [torch.zeros(n * 12, dtype=torch.uint8, device=dev) for _ in range(2)]
pr = torch.zeros(n * 8, dtype=torch.uint8, device=dev)
cap = 1 << 22
corr = torch.zeros(cap * 8, dtype=torch.uint8, device=dev); nc = torch.zeros(1, dtype=torch.int32, device=dev)
b = abi.Batch(); b.n, b.flags = n, abi.BATCH_STAT_ISIZE
b.seq1, b.qual1, b.len1 = (mates[0][k].data_ptr() for k in ("seq", "qual", "lens"))
b.seq2, b.qual2, b.len2 = (mates[1][k].data_ptr() for k in ("seq", "qual", "lens"))
r = abi.Results(); r.r1, r.r2, r.pair = res[0].data_ptr(), res[1].data_ptr(), pr.data_ptr()
r.corrections, r.corrections_capacity, r.n_corrections = corr.data_ptr(), cap, nc.data_ptr()
torch.cuda.synchronize()
g.submit_device(b, r); g.synchronize()
fin = []
for m in range(2):
    f = abi.FormatIn(); f.text, f.line_off, f.line_len, f.res = mates[m]["t"].data_ptr(), mates[m]["loff"].data_ptr(), mates[m]["llen"].data_ptr(), res[m].data_ptr()
    fin.append(f)
outs = [torch.empty(mates[m]["nbytes"] + 16, dtype=torch.uint8, device=dev) for m in range(2)]
torch.cuda.synchronize()
for it in range(4):
    t0 = time.perf_counter()
    rc, l1, l2 = g.format_fastq(n, fin[0], fin[1], corr.data_ptr(), nc.data_ptr(), outs[0].data_ptr(), outs[0].numel(), outs[1].data_ptr(), outs[1].numel())
    dt = time.perf_counter() - t0
    print(f"{n} pairs, {int(nc.item())} corrections, {(l1+l2)/1e6:.1f} MB out in {dt*1e3:.2f} ms -> {2*n/dt/1e6:.1f} Mreads/s, {(l1+l2)/dt/1e9:.2f} GB/s written")
