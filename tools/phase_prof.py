"""Per-phase cycle breakdown of the fused kernel on the bench workload (debug aid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests'); sys.path.insert(0, ROOT + '/tools')
os.environ["FASTP_GPU_PHASE_TIMING"] = "1"
import numpy as np, torch
from fastp_amd import abi, engine
import synth_torch
import bench
dev = torch.device('cuda', 0)
p, _ = bench.bench_params()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2 * 1024 * 1024
d = synth_torch.synth_pairs_torch(n, L=150, seed=42, device=dev)
s1, q1, l1 = synth_torch.pack_torch(d['seq1'], d['qual1'], d['len1'], 150)
s2, q2, l2 = synth_torch.pack_torch(d['seq2'], d['qual2'], d['len2'], 150)
del d
g = engine.GpuEngine(p)
r1 = torch.zeros(n * 12, dtype=torch.uint8, device=dev); r2 = torch.zeros(n * 12, dtype=torch.uint8, device=dev)
pr = torch.zeros(n * 8, dtype=torch.uint8, device=dev); nc = torch.zeros(1, dtype=torch.int32, device=dev)
b = abi.Batch(); b.n = n; b.flags = 1
b.seq1, b.qual1, b.len1 = s1.data_ptr(), q1.data_ptr(), l1.data_ptr()
b.seq2, b.qual2, b.len2 = s2.data_ptr(), q2.data_ptr(), l2.data_ptr()
res = abi.Results(); res.r1, res.r2, res.pair = r1.data_ptr(), r2.data_ptr(), pr.data_ptr()
res.corrections = None; res.corrections_capacity = 0; res.n_corrections = nc.data_ptr()
torch.cuda.synchronize()
g.submit_device(b, res); g.synchronize(); g.kernel_time(); g.debug_phase_cycles()
for it in range(2):
    g.submit_device(b, res); g.synchronize()
    ms, k = g.kernel_time()
    cyc = g.debug_phase_cycles()
    tot = sum(cyc[:10])
    names = ["load", "hash", "trim", "polyg", "overlap", "decide", "filter", "stats", "masks", "metrics"]
    print(f"kernel {ms:.3f} ms / {k} launches; phase share:", {nm: f"{100.0 * c / tot:.1f}%" for nm, c in zip(names, cyc)})
