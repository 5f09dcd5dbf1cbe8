"""One step of the worker loop as a timeline: every kernel of a steady-state step with its start (from the step's lane kernel's
start), its duration and its queue, from a rocprofv3 --kernel-trace of the bench - shows what the step's wall clock holds besides the
two kernels of the roofline line (Duplicate's tail on the second stream, the slab folds, the joins between the streams).
    python tools/step_timeline.py gpurun_out/prof/<dir> [steps to print]   (the directory of a run WITH its *_kernel_trace.csv)"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    show = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        print("no kernel trace under", d)
        return
    rows = []
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"]
        if "fq_" not in name and "rocclr" not in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0].replace("void ", "")[:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "fq_lane_kernel" in r[2]]
    if len(starts) < show + 2:
        print("too few steps in the trace")
        return
    periods = [rows[starts[i + 1]][0] - rows[starts[i]][0] for i in range(len(starts) - 1)]
    mid = sorted(periods)[len(periods) // 2]
    print(f"{len(starts)} steps; lane kernel start to next lane kernel start: median {mid / 1e3:.1f} us, min {min(periods) / 1e3:.1f}, max {max(periods) / 1e3:.1f}")
    # steady-state steps: the last ones whose period is within 5 % of the median
    picked = [i for i in range(len(starts) - 1) if abs(periods[i] - mid) < 0.05 * mid][-show:]
    for i in picked:
        t0 = rows[starts[i]][0]
        print(f"-- step {i}: {periods[i] / 1e3:.1f} us")
        busy_end = t0
        for s, e, name, q, st in rows[starts[i]:starts[i + 1]]:
            gap = (s - busy_end) / 1e3
            print(f"   +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  queue {q:>3s} stream {st:>3s}  {name}" + (f"   (nothing ran for {gap:.1f} us before it)" if gap > 2.0 else ""))
            busy_end = max(busy_end, e)
        print(f"   +{(busy_end - t0) / 1e3:8.1f} us  last kernel ends; next lane kernel starts {(rows[starts[i + 1]][0] - busy_end) / 1e3:.1f} us later")


if __name__ == "__main__":
    main()
