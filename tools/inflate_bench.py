"""Throughput of fastp_gpu_inflate_bgzf: BGZF-compressed synthetic FASTQ text (zlib level 6, 0xff00-byte blocks) in HBM."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests'); sys.path.insert(0, ROOT + '/tools')
import numpy as np, torch
from fastp_amd import abi, engine
import synth_torch, bgzf_util
from concurrent.futures import ThreadPoolExecutor
dev = torch.device('cuda', 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
strategy = {"default": zlib.Z_DEFAULT_STRATEGY, "huffman": zlib.Z_HUFFMAN_ONLY, "fixed": zlib.Z_FIXED, "rle": zlib.Z_RLE}[sys.argv[2] if len(sys.argv) > 2 else "default"]
level = int(sys.argv[3]) if len(sys.argv) > 3 else 6
d = synth_torch.synth_pairs_torch(200_000, L=150, seed=3, device=dev)
text = synth_torch.to_fastq_tensor(d["seq1"], d["qual1"], 1).cpu().numpy().tobytes()
t0 = time.time()
with ThreadPoolExecutor(32) as pool:   # zlib releases the GIL
    blocks = list(pool.map(lambda i: bgzf_util.block(text[i:i + 0xff00], level, strategy), range(0, len(text), 0xff00)))
one = b"".join(blocks)
print(f"{len(text)/1e6:.1f} MB text -> {len(one)/1e6:.1f} MB BGZF ({len(blocks)} blocks, ratio {len(text)/len(one):.2f}) in {time.time()-t0:.1f}s", flush=True)
g = engine.GpuEngine(abi.default_params(False, 150))
for r in (1, reps):
    comp = one * r
    host = np.frombuffer(comp, dtype=np.uint8)
    t0 = time.perf_counter()
    info, poff, plen, isz, crc, ooff = g.bgzf_index(host, len(blocks) * r + 8, 1 << 40)
    t_idx = time.perf_counter() - t0
    d_comp = torch.frombuffer(bytearray(comp + b"\0" * 16), dtype=torch.uint8).to(dev)
    arrs = [torch.from_numpy(a.copy()).to(dev) for a in (poff, plen, isz, crc, ooff)]
    out = torch.empty(int(info.out_bytes) + 16, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for crc_on in (True, False):
        for it in range(2):
            t0 = time.perf_counter()
            g.inflate_bgzf(d_comp.data_ptr(), info.n_blocks, *[a.data_ptr() for a in arrs], out.data_ptr(), int(info.out_bytes), crc_on)
            dt = time.perf_counter() - t0
        print(f"{info.n_blocks} blocks, {info.out_bytes/1e6:.0f} MB text, crc={crc_on}: {dt*1e3:.2f} ms -> {info.out_bytes/dt/1e9:.2f} GB/s of text "
              f"({len(comp)/dt/1e9:.2f} GB/s compressed); host index walk {t_idx*1e3:.2f} ms", flush=True)
    ok = out[:len(text)].cpu().numpy().tobytes() == text
    print("first copy identical to the source text:", ok, flush=True)
