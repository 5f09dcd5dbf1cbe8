#!/usr/bin/env python3
"""What units with letters outside ACGTN cost: the engine on 400 k synthetic 2x150 pairs (host buffers: PCIe inside) with a
growing share of such units; kernel time from the engine's own events (plan's kernels + the text kernel)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from fastp_amd import abi, engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
p = abi.default_params(True, 150)
p.cut_right = 1
base = synth.synth_pairs(n, L=150, seed=5, insert_mean=220.0)
print(f"{n} pairs 2x150, default options + --cut_right; share = reads (per mate) given letters outside ACGTN")
for share in (0.0, 0.0001, 0.001, 0.01, 0.1, 1.0):
    d = {k: v.copy() for k, v in base.items()}
    if share > 0:
        synth.add_exotic(d, seed=3, read_frac=share)
    eng = engine.GpuEngine(p, device=0)
    eng.process(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])   # warm-up (allocations)
    eng.reset()
    eng.kernel_time()
    t0 = time.perf_counter()
    eng.process(d["seq1"], d["qual1"], d["len1"], d["seq2"], d["qual2"], d["len2"])
    wall = time.perf_counter() - t0
    ms, launches = eng.kernel_time()
    units = int(((d["seq1"] > 96) | (d["seq1"] == 46) | (d["seq2"] > 96) | (d["seq2"] == 46)).any(axis=1).sum()) if share else 0
    print(f"share {share:7.4f}: ~{units:7d} units for the text kernel; kernels {ms:9.3f} ms ({2 * n / ms / 1e3:9.1f} Mreads/s), process() wall {wall * 1e3:8.1f} ms")
    eng.close()
