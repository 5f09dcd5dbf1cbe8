"""Does the patched reference's wall clock depend on where its threads run?  The 12 M-pair drop-in run under taskset on each NUMA
node's CPUs (and unpinned), fresh outputs, 3 runs each: wall, the loop's own time.   python tools/numa_probe.py [PAIRS]"""
import glob
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


def main():
    import torch
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
    dev = torch.device("cuda", 0)
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        nodes[os.path.basename(d)] = open(os.path.join(d, "cpulist")).read().strip()
    print("NUMA nodes:", nodes)
    for f in sorted(glob.glob("/sys/class/drm/card*/device/local_cpulist")):
        ven = open(os.path.join(os.path.dirname(f), "vendor")).read().strip()
        node = open(os.path.join(os.path.dirname(f), "numa_node")).read().strip()
        print(f"{f}: vendor {ven} numa_node {node} local_cpulist {open(f).read().strip()}")
    print("this process may run on:", sorted(os.sched_getaffinity(0))[:4], "...", len(os.sched_getaffinity(0)), "CPUs", flush=True)
    tmp, f1, f2 = bench.write_sample_files(pairs, dev)
    refgpu = os.path.join(ROOT, "oracle", "_ref", "fastp_ref_gpu")
    cmd = [refgpu, "-i", f1, "-I", f2, "-o", os.path.join(tmp, "d1.fq"), "-O", os.path.join(tmp, "d2.fq"), "-j", os.path.join(tmp, "d.json"),
           "-h", os.path.join(tmp, "d.html"), "-w", "16"] + bench.bench_params()[1]
    env = dict(os.environ, FASTP_GPU="1", FASTP_GPU_VERBOSE="1")
    for name, pre in [("unpinned", [])] + [(f"taskset {n} ({c})", ["taskset", "-c", c]) for n, c in nodes.items()] + [("unpinned again", [])]:
        walls, loops = [], []
        for _ in range(3):
            bench.fresh_outputs(cmd)
            t0 = time.time()
            pr = subprocess.run(pre + cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600, env=env)
            walls.append(time.time() - t0)
            m = re.search(r"chunks, ([0-9.]+) s \(setup", pr.stderr.decode(errors="replace"))
            loops.append(float(m.group(1)) if m else -1)
        print(f"{name:40s} wall {' '.join(f'{w * 1e3:6.0f}' for w in walls)} ms   loop {' '.join(f'{x * 1e3:5.0f}' for x in loops)} ms", flush=True)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
