#!/usr/bin/env python3
"""Feasibility probe: do the lane kernel of one batch and the Stats kernel of another run faster side by side than one
after the other?  Two engine contexts (each with its own stream) take alternating batches from two host threads; the GPU's
dispatcher co-schedules whatever fits.  (Two contexts = two bloom filters: the duplicate semantics are NOT those of one
run; this only measures the hardware's behaviour.)    python tools/two_engine_probe.py [engines] [steps]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import bench  # noqa: E402
from fastp_amd import abi, engine  # noqa: E402

NE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 32
B = 4 * 1024 * 1024
dev = torch.device("cuda", 0)
params, _ = bench.bench_params()
engs = [engine.GpuEngine(params, device=0) for _ in range(NE)]
batches = [bench.ResidentBatch(B, 4242 + i, dev) for i in range(4)]


def results():
    r = abi.Results()
    t = [torch.zeros(B * 12, dtype=torch.uint8, device=dev), torch.zeros(B * 12, dtype=torch.uint8, device=dev),
         torch.zeros(B * 8, dtype=torch.uint8, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)]
    r.r1, r.r2, r.pair = t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr()
    r.corrections, r.corrections_capacity, r.n_corrections = None, 0, t[3].data_ptr()
    return r, t


res = [results() for _ in range(NE)]
torch.cuda.synchronize(dev)


def work(k, steps):
    e = engs[k]
    for s in range(steps):
        e.submit_device(batches[(s * NE + k) % len(batches)].batch, res[k][0])
    e.synchronize()


for k in range(NE):
    work(k, 2)
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
ths = [threading.Thread(target=work, args=(k, STEPS // NE)) for k in range(NE)]
for t in ths:
    t.start()
for t in ths:
    t.join()
torch.cuda.synchronize(dev)
dt = time.perf_counter() - t0
n = (STEPS // NE) * NE
print(f"engines {NE} plan {engs[0].plan()} steps {n}: {dt / n * 1e3:.3f} ms per 4 Mi-pair batch, {2 * B * n / dt / 1e6:.1f} Mreads/s "
      f"(LANE_BLOCKS_PER_CU={os.environ.get('FASTP_GPU_LANE_BLOCKS_PER_CU', '-')} STATS_THREADS={os.environ.get('FASTP_GPU_STATS_THREADS', '-')} "
      f"STATS_BLOCKS_PER_CU={os.environ.get('FASTP_GPU_STATS_BLOCKS_PER_CU', '-')})", flush=True)
