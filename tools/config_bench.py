"""Engine throughput on the other BASELINE.json configurations (inputs resident in HBM, like bench.py; these are not
bench lines): configs[1] SE 1x150, 10 M reads, sliding-window quality trim + polyG; configs[4]'s per-GPU share:
PE 2x250 with --dedup and the overrepresentation analysis."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests'); sys.path.insert(0, ROOT + '/tools')
import numpy as np, torch
from fastp_amd import abi, engine, hostloop
import synth_torch, cases
import evalport

dev = torch.device('cuda', 0)


def run(name, params, L, n, paired, steps=5):
    d = synth_torch.synth_pairs_torch(n, L=L, seed=5, device=dev)
    bufs = {}
    for m in ("1", "2") if paired else ("1",):
        bufs[m] = synth_torch.pack_torch(d["seq" + m], d["qual" + m], d["len" + m], L)
    del d
    eng = engine.GpuEngine(params)
    r1 = torch.zeros(n * 12, dtype=torch.uint8, device=dev); r2 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
    pr = torch.zeros(n * 8, dtype=torch.uint8, device=dev); nc = torch.zeros(1, dtype=torch.int32, device=dev)
    b = abi.Batch(); b.n, b.flags = n, abi.BATCH_STAT_ISIZE
    b.seq1, b.qual1, b.len1 = (x.data_ptr() for x in bufs["1"])
    if paired:
        b.seq2, b.qual2, b.len2 = (x.data_ptr() for x in bufs["2"])
    r = abi.Results(); r.r1 = r1.data_ptr()
    if paired:
        r.r2, r.pair = r2.data_ptr(), pr.data_ptr()
    r.n_corrections = nc.data_ptr()
    torch.cuda.synchronize()
    eng.submit_device(b, r); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.submit_device(b, r)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / steps
    reads = n * (2 if paired else 1)
    kms, kl = eng.kernel_time()
    print(f"{name}: {n} {'pairs' if paired else 'reads'} of {L} bp per step, {dt*1e3:.2f} ms/step -> {reads/dt/1e6:.0f} Mreads/s "
          f"(fused kernel {kms/max(1,kl):.2f} ms per launch, {kl//(steps+1)} launches per step)", flush=True)
    eng.close()


p = abi.default_params(False, 150)
p.adapter_seq_r1 = None; p.adapter_enabled = 0; p.poly_g = 1; p.cut_right = 1
run("configs[1] SE 1x150, -A -g --cut_right", p, 150, 10_000_000, False)

L = 250
p = abi.default_params(True, L); p.cut_right = 1; p.dedup = 1
n = 2_000_000
d = synth_torch.synth_pairs_torch(20000, L=L, seed=5, device="cpu")
pad = lambda a: np.pad(a.numpy(), ((0, 0), (0, 6)))
b1 = cases._ArrayBatch(pad(d["seq1"]), d["len1"].numpy()); b2 = cases._ArrayBatch(pad(d["seq2"]), d["len2"].numpy())
e1, e2 = evalport.evaluate_seq_len(b1), evalport.evaluate_seq_len(b2)
abi.set_overrep(p, evalport.evaluate_overrep_seqs(b1, e1), evalport.evaluate_overrep_seqs(b2, e2), e1, e2, 20)
run(f"configs[4] share: PE 2x250, --dedup, -p ({p.n_overrep_seqs1}+{p.n_overrep_seqs2} seeds)", p, L, n, True)
