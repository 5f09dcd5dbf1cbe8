"""Where the patched reference's wall clock goes (FASTP_GPU=1 fastp_ref_gpu, the stream binding): the start-up timeline
(FASTP_GPU_TIMELINE=1, fq_timeline.h) on 1000 pairs and on the whole sample, and the file loop under a few I/O settings.
    python tools/dropin_probe.py PAIRS [PAIRS ...] > gpurun_out/<visit>_dropin_probe.txt
Each configuration: best of 3 whole-process walls + the loop's own breakdown line (FASTP_GPU_VERBOSE=1)."""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def run(cmd, env, reps=3):
    best, err = None, ""
    outs = [cmd[i + 1] for i, a in enumerate(cmd) if a in ("-o", "-O")]
    for _ in range(reps):
        for f in outs:   # fresh outputs every time: truncating a file of gigabytes on tmpfs (O_TRUNC) costs 0.1 - 0.5 s of its own
            if os.path.exists(f):
                os.remove(f)
        t0 = time.time()
        pr = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, check=True, timeout=900, env=env)
        dt = time.time() - t0
        if best is None or dt < best:
            best, err = dt, pr.stderr.decode(errors="replace")
    return best, err


def main():
    import torch
    dev = torch.device("cuda", 0)
    refgpu = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu")
    flags = bench.bench_params()[1]
    for pairs in [int(a) for a in sys.argv[1:]] or [4_000_000]:
        tmp, f1, f2 = bench.write_sample_files(pairs, dev)
        cmd = [refgpu, "-i", f1, "-I", f2, "-o", os.path.join(tmp, "d1.fq"), "-O", os.path.join(tmp, "d2.fq"), "-j", os.path.join(tmp, "d.json"),
               "-h", os.path.join(tmp, "d.html"), "-w", str(bench.host_cores())] + flags
        base = dict(os.environ, FASTP_GPU="1", FASTP_GPU_VERBOSE="1")
        print(f"==== {pairs} pairs of 2x150, plain FASTQ on tmpfs ({tmp})", flush=True)
        # the timelines: the first 1000 pairs, the whole sample
        for what, extra in (("first 1000 pairs", ["--reads_to_process", "1000"]), ("whole sample", [])):
            wall, err = run(cmd + extra, dict(base, FASTP_GPU_TIMELINE="1"), reps=2)
            print(f"-- timeline, {what}: wall {wall * 1e3:.0f} ms")
            for ln in err.splitlines():
                if "timeline" in ln or "stream mode:" in ln:
                    print("   " + ln.strip()[:330])
            sys.stdout.flush()
        # the same binary without the engine on 1000 pairs: what the reference's own start and reports cost
        wall, _ = run(cmd + ["--reads_to_process", "1000"], dict(os.environ, FASTP_GPU="0"), reps=2)
        print(f"-- FASTP_GPU=0, first 1000 pairs: wall {wall * 1e3:.0f} ms")
        # the file loop under a few settings
        for name, env in (("defaults", {}),
                          ("orderly exit (FASTP_GPU_FAST_EXIT=0)", {"FASTP_GPU_FAST_EXIT": "0"}),
                          ("read pieces 8 MiB (round 5)", {"FASTP_GPU_STREAM_READ_PIECE_KB": "8192"}),
                          ("defaults again", {})) + tuple(
                              (a, dict(kv.split("=", 1) for kv in a.split())) for a in os.environ.get("PROBE_EXTRA", "").split(";") if a):
            wall, err = run(cmd, dict(base, **env))
            m = re.search(r"stream mode: .*", err)
            print(f"-- {name:38s} wall {wall * 1e3:6.0f} ms = {2 * pairs / wall / 1e6:6.2f} Mreads/s   {m.group(0)[13:300] if m else ''}", flush=True)
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)


if __name__ == "__main__":
    main()
