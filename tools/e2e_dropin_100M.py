"""BASELINE configs[2] at its full size through the DROP-IN: 100 M synthetic 2x150 pairs as plain FASTQ on tmpfs ->
`FASTP_GPU=1 fastp_ref_gpu` (the reference with the stream binding) next to `fastp_ref -w 1` (the semantics the engine
implements) on the same files: out1 / out2 compared byte for byte, the two JSON reports - both written by the reference's own
JsonReporter - value for value, nothing excepted.
usage: python tools/e2e_dropin_100M.py [--pairs N] [--ref-threads 1]"""
import argparse, json, os, re, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tools")
import torch
import synth_torch

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=100_000_000)
ap.add_argument("--ref-threads", type=int, default=1)
ap.add_argument("--no-ref", action="store_true", help="time the drop-in only (several stream settings), no reference run")
ap.add_argument("--also", type=int, default=16, help="a second, timing-only run of the reference with this many threads (0 = none)")
args = ap.parse_args()
L = 150
dev = torch.device("cuda", 0)
base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
tmp = tempfile.mkdtemp(prefix="fastp_dropin_", dir=base)
f1, f2 = tmp + "/r1.fq", tmp + "/r2.fq"
t0 = time.time()
block = 1_000_000
with open(f1, "wb", buffering=0) as a, open(f2, "wb", buffering=0) as b:
    for done in range(0, args.pairs, block):
        k = min(block, args.pairs - done)
        d = synth_torch.synth_pairs_torch(k, L=L, seed=1000 + done // block, device=dev)   # distinct fragments per block
        for mate, fh in ((1, a), (2, b)):
            rec = synth_torch.to_fastq_tensor(d[f"seq{mate}"], d[f"qual{mate}"], mate, first=done).cpu().numpy()
            fh.write(memoryview(rec).cast("B"))
        del d
torch.cuda.empty_cache()
nbytes = os.path.getsize(f1) + os.path.getsize(f2)
print(f"input: {args.pairs} pairs 2x{L} bp, {nbytes} bytes of plain FASTQ on {base}, generated in {time.time()-t0:.1f}s; host has {os.cpu_count()} logical cores", flush=True)
flags = ["-G", "--cut_right"]


def run(binary, tag, w, env=None):
    cmd = [binary, "-i", f1, "-I", f2, "-o", f"{tmp}/{tag}1.fq", "-O", f"{tmp}/{tag}2.fq", "-j", f"{tmp}/{tag}.json", "-h", f"{tmp}/{tag}.html", "-w", str(w)] + flags
    for k in "12":   # an earlier run's outputs: truncating ~30 GB of tmpfs inside open() would be timed (3 - 4 s of the A/B runs of round 4)
        if os.path.exists(f"{tmp}/{tag}{k}.fq"):
            os.unlink(f"{tmp}/{tag}{k}.fq")
    t0 = time.time()
    p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})), timeout=1500)
    dt = time.time() - t0
    if p.returncode != 0:
        raise SystemExit(f"{binary} failed: {p.stderr.decode()[-800:]}")
    rep = json.load(open(f"{tmp}/{tag}.json"))
    rep.pop("command", None)
    return dt, rep, p.stderr.decode(errors="replace")


if args.no_ref:
    for label, env in (("default", {}), ("read threads 32", {"FASTP_GPU_STREAM_READ_THREADS": "32"}), ("read threads 8", {"FASTP_GPU_STREAM_READ_THREADS": "8"}),
                       ("chunks of 32 MiB", {"FASTP_GPU_STREAM_CHUNK_MB": "32"})):
        tg, rg, err = run(os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu"), "g", 16, dict({"FASTP_GPU": "1", "FASTP_GPU_VERBOSE": "1"}, **env))
        m = re.search(r"fastp_gpu: stream mode: .*", err)
        print(f"FASTP_GPU=1 fastp_ref_gpu -w 16, {label}: {tg:.2f} s = {2*args.pairs/tg/1e6:.2f} Mreads/s\n   " + (m.group(0) if m else ""), flush=True)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    raise SystemExit(0)
tg, rg, err = run(os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu"), "g", 16, {"FASTP_GPU": "1", "FASTP_GPU_VERBOSE": "1"})
print(f"FASTP_GPU=1 fastp_ref_gpu -w 16 (stream binding): {tg:.2f} s = {2*args.pairs/tg/1e6:.2f} Mreads/s", flush=True)
m = re.search(r"fastp_gpu: stream mode: .*", err)
print("   " + (m.group(0) if m else "(no stream line)"), flush=True)
tr, rr, _ = run(os.path.join(ROOT, "oracle", "_ref", "fastp_ref"), "o", args.ref_threads)
print(f"fastp_ref -w {args.ref_threads}: {tr:.2f} s = {2*args.pairs/tr/1e6:.2f} Mreads/s", flush=True)


def same(pa, pb):
    if os.path.getsize(pa) != os.path.getsize(pb):
        return False
    n = os.path.getsize(pa)
    step = 256 << 20

    def part(off):
        with open(pa, "rb") as x, open(pb, "rb") as y:
            x.seek(off); y.seek(off)
            return x.read(step) == y.read(step)
    with ThreadPoolExecutor(32) as ex:
        return all(ex.map(part, range(0, n, step)))


t0 = time.time()
eq = [same(f"{tmp}/o{k}.fq", f"{tmp}/g{k}.fq") for k in (1, 2)]
print(f"out1 / out2 byte-identical to fastp_ref -w {args.ref_threads}'s files: {eq} ({os.path.getsize(tmp + '/g1.fq')} + {os.path.getsize(tmp + '/g2.fq')} bytes, compared in {time.time()-t0:.1f}s)", flush=True)


def diff(x, y, path, out):
    if isinstance(x, dict) and isinstance(y, dict):
        for k in sorted(set(x) | set(y)):
            if k not in x or k not in y:
                out.append(f"{path}/{k}: only on one side")
            else:
                diff(x[k], y[k], f"{path}/{k}", out)
    elif isinstance(x, list) and isinstance(y, list):
        if len(x) != len(y):
            out.append(f"{path}: {len(x)} vs {len(y)} entries")
        else:
            for i, (p, q) in enumerate(zip(x, y)):
                diff(p, q, f"{path}[{i}]", out)
    elif x != y:
        out.append(f"{path}: reference {x!r} binding {y!r}")


def leaves(x):
    return sum(leaves(v) for v in (x.values() if isinstance(x, dict) else x)) if isinstance(x, (dict, list)) else 1


problems = []
diff(rr, rg, "", problems)
print(f"JSON report (the reference's own JsonReporter on both sides, {leaves(rr)} values, nothing excepted) vs fastp_ref -w {args.ref_threads}: "
      f"{'IDENTICAL' if not problems else str(len(problems)) + ' values differ: ' + '; '.join(problems[:8])}", flush=True)
print(f"summary: total_reads {rr['summary']['before_filtering']['total_reads']}, passed {rr['filtering_result']['passed_filter_reads']}, "
      f"duplication rate {rr['duplication']['rate']}, adapter-trimmed reads {rr['adapter_cutting']['adapter_trimmed_reads']}, insert size peak {rr['insert_size']['peak']}", flush=True)
print(f"speedup end to end vs fastp_ref -w {args.ref_threads}: {tr/tg:.1f}x", flush=True)
if args.also:
    for k in (1, 2):
        os.unlink(f"{tmp}/o{k}.fq")
    t2, r2, _ = run(os.path.join(ROOT, "oracle", "_ref", "fastp_ref"), "o", args.also)
    print(f"fastp_ref -w {args.also} (its fastest setting on this box): {t2:.2f} s = {2*args.pairs/t2/1e6:.2f} Mreads/s -> drop-in speedup {t2/tg:.1f}x; "
          f"its JSON differs from its own -w {args.ref_threads} run in sections {[k for k in rr if rr[k] != r2.get(k)]}", flush=True)
import shutil
shutil.rmtree(tmp, ignore_errors=True)
