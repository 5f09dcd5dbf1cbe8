"""timings of the steps either side of the path on one GPU: formatter (all streams), gzip members, Evaluator
pre-pass.  python tools/aux_bench.py [n_pairs]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
import engines, format_util, synth, cases
from fastp_amd import abi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
mem = format_util.TorchMem()
paired, flags, pf, skw = cases.CASES["pe_filters"]
d = synth.synth_pairs(n, L=150, seed=1)
params = cases.finalize_params("pe_filters", pf(150), d["seq1"], d["len1"], d["seq2"], d["len2"])
fq1 = synth.to_fastq(d["seq1"], d["qual1"], d["len1"], 1)
fq2 = synth.to_fastq(d["seq2"], d["qual2"], d["len2"], 2)
g = engines.gpu_engine(params)

def timed(label, fn, nbytes, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label}: {dt * 1e3:.2f} ms  {nbytes / dt / 1e9:.2f} GB/s ({nbytes / 1e6:.1f} MB)", flush=True)
    return dt

# formatter: all streams
c = format_util._prepare(g, mem, fq1, fq2, 150)
ios = []
for m in range(2):
    f = abi.FormatIn()
    f.text, f.line_off, f.line_len, f.res = (mem.ptr(c["mates"][m]["text"]), mem.ptr(c["mates"][m]["loff"]),
                                             mem.ptr(c["mates"][m]["llen"]), mem.ptr(c["res"][m]))
    ios.append(f)
o = abi.FormatOptions()
o.want_failed = o.want_unpaired1 = o.want_unpaired2 = 1
cap = len(fq1) + len(fq2) + n * 64
outs = [mem.alloc(cap) for _ in range(6)]
lens_box = {}
def fmt():
    rc, lens = g.format_streams(c["n"], ios[0], ios[1], mem.ptr(c["pair"]), mem.ptr(c["corr"]), mem.ptr(c["nc"]), o,
                                [mem.ptr(x) for x in outs], [cap] * 6)
    lens_box["l"] = lens
timed("format_streams (6 streams)", fmt, len(fq1) + len(fq2))
lens = lens_box["l"]
print("stream bytes", lens)
# gzip members of out1
comp = mem.alloc(lens[0] + 31 * (lens[0] // 65280 + 1) + 64)
box = {}
def defl():
    rc, nb = g.deflate_bgzf(mem.ptr(outs[0]), lens[0], mem.ptr(comp), comp.numel(), True)
    box["n"] = nb
timed("deflate_bgzf(out1)", defl, lens[0])
print(f"ratio {box['n'] / lens[0]:.4f}")
import zlib
txt = mem.download(outs[0], min(lens[0], 20 << 20))
t0 = time.perf_counter(); z = zlib.compress(txt, 4); dt = time.perf_counter() - t0
print(f"zlib level 4 on one host core: {len(txt) / dt / 1e6:.1f} MB/s, ratio {len(z) / len(txt):.4f}")
# inflate it back (the existing one-lane-per-block inflate)
import test_hostsim_parity as hs
cbytes = mem.download(comp, box["n"])
host = np.frombuffer(cbytes, dtype=np.uint8)
info, poff, plen, isz, crc, ooff = g.bgzf_index(host, 1 << 22, 1 << 40)
d_comp = mem.upload(cbytes, 16)
dev = [mem.upload(a.tobytes(), 16) for a in (poff, plen, isz, crc, ooff)]
back = mem.alloc(int(info.out_bytes))
def infl():
    g.inflate_bgzf(mem.ptr(d_comp), info.n_blocks, *[mem.ptr(x) for x in dev], mem.ptr(back), int(info.out_bytes), 1)
timed("inflate_bgzf", infl, int(info.out_bytes))
assert mem.download(back, 1 << 20) == mem.download(outs[0], 1 << 20)
# Evaluator
m0 = c["mates"][0]
counts = mem.alloc(4 << 20)
timed("eval_adapter_kmers", lambda: g.eval_adapter_kmers(mem.ptr(m0["seq"]), mem.ptr(m0["qual"]), mem.ptr(m0["lens"]), c["n"], 0, mem.ptr(counts)),
      min(c["n"], 256 * 1024) * 150)
sl = g.eval_seq_len(mem.ptr(m0["lens"]), c["n"])
timed("eval_overrep", lambda: g.eval_overrep(mem.ptr(m0["seq"]), mem.ptr(m0["qual"]), mem.ptr(m0["lens"]), c["n"], sl), 1510000)
g.close()
