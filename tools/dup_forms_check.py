"""the two forms of the Duplicate kernels (probe/resolve table form vs claim/winners/finish, in-kernel claim) on the same
stream of batches: every counter and every RF_DUP flag must be identical.  python tools/dup_forms_check.py [batches] [pairs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tools")
import numpy as np, torch
from fastp_amd import abi, engine
import synth_torch

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4 * 1024 * 1024
L = 150
dev = torch.device("cuda", 0)
p = abi.default_params(True, L); p.cut_right = 1
batches = []
for k in range(NB):
    d = synth_torch.synth_pairs_torch(B, L=L, seed=7000 + k, device=dev)
    if k == NB - 1:   # the last batch repeats half of the first one: cross-batch duplicates, and same-launch ones inside
        d0 = synth_torch.synth_pairs_torch(B, L=L, seed=7000, device=dev)
        for key in d:
            d[key][: B // 2] = d0[key][: B // 2]
            d[key][B // 2: B // 2 + B // 8] = d[key][B // 2 + B // 8: B // 2 + B // 4]
    t = synth_torch.pack_torch(d["seq1"], d["qual1"], d["len1"], L) + synth_torch.pack_torch(d["seq2"], d["qual2"], d["len2"], L)
    batches.append(t)
    del d

def run(env):
    for k, v in env.items():
        os.environ[k] = v
    eng = engine.GpuEngine(p, device=0)
    flags = []
    r1 = torch.zeros(B * 12, dtype=torch.uint8, device=dev); r2 = torch.zeros_like(r1); pr = torch.zeros(B * 8, dtype=torch.uint8, device=dev)
    nc = torch.zeros(1, dtype=torch.int32, device=dev)
    for t in batches:
        b = abi.Batch(); b.n, b.flags = B, abi.BATCH_STAT_ISIZE
        b.seq1, b.qual1, b.len1 = (x.data_ptr() for x in t[:3]); b.seq2, b.qual2, b.len2 = (x.data_ptr() for x in t[3:])
        res = abi.Results(); res.r1, res.r2, res.pair = r1.data_ptr(), r2.data_ptr(), pr.data_ptr(); res.n_corrections = nc.data_ptr()
        eng.submit_device(b, res); eng.synchronize()
        flags.append((r1.view(-1, 12)[:, 5] & abi.RF_DUP).cpu().numpy().copy())
    ctr = eng.counters().copy()
    eng.close()
    for k in env:
        os.environ.pop(k, None)
    return ctr, flags

c_new, f_new = run({})
c_tab, f_tab = run({"FASTP_GPU_DUP_TABLE": "1"})
c_ker, f_ker = run({"FASTP_GPU_CLAIM_FUSED": "0"})
print("duplicates flagged per batch (in-kernel claim):", [int((f != 0).sum()) for f in f_new])
for name, c, f in (("table form", c_tab, f_tab), ("claim kernel", c_ker, f_ker)):
    same_c = np.array_equal(c_new, c)
    same_f = all(np.array_equal(a, b) for a, b in zip(f_new, f))
    print(f"in-kernel claim vs {name}: counters identical {same_c}, RF_DUP flags identical {same_f}")
    assert same_c and same_f
