#!/usr/bin/env python3
"""Condense one GPU-box visit's rocprofv3 output (gpurun_out/prof/<tag>, <tag>_fetch, <tag>_write)
into profiles/<tag>_*.  HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream -> doubled
(checked here against the known input footprint of the fused kernel); WRITE_SIZE is used as is."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
PAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 4194304   # pairs per launch of the profiled runs (bench.py --batches 2: 4 Mi)
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.environ.get("SUMMARIZE_DST") or os.path.join(ROOT, "profiles")   # (on the GPU box: a directory under gpurun_out/, which is what travels back)
os.makedirs(dst, exist_ok=True)

stats = list(csv.DictReader(open(os.path.join(src, tag, "trace_kernel_stats.csv"))))
ours = [r for r in stats if r["Name"].startswith("fq_") or "fq_lane_kernel" in r["Name"]]
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
    w.writeheader()
    for r in stats:
        r = dict(r)
        if len(r["Name"]) > 120:
            r["Name"] = r["Name"][:117] + "..."
        w.writerow(r)

tp = os.path.join(src, tag, "trace_kernel_trace.csv")
trace = list(csv.DictReader(open(tp))) if os.path.exists(tp) else []
geom = {}
for r in trace:
    if (r["Kernel_Name"].startswith("fq_") or "fq_lane_kernel" in r["Kernel_Name"]) and r["Kernel_Name"] not in geom:
        geom[r["Kernel_Name"]] = {k: r[k] for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count",
                                                    "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X")}

pmc = {}
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    p = os.path.join(src, f"{tag}_{kind}", "pmc_counter_collection.csv")
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        if (r["Kernel_Name"].startswith("fq_") or "fq_lane_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc.setdefault(k, {})[counter + "_KiB_avg"] = sum(v) / len(v)
        pmc[k][counter + "_launches"] = len(v)

out = {"tag": tag, "kernels": {}}
for r in ours:
    k = r["Name"]
    e = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6,
         "max_ms": float(r["MaxNs"]) / 1e6}
    e.update(geom.get(k, {}))
    if k in pmc:
        e.update(pmc[k])
        fetch = pmc[k].get("FETCH_SIZE_KiB_avg")
        write = pmc[k].get("WRITE_SIZE_KiB_avg")
        if fetch is not None and write is not None:
            e["hbm_read_bytes_per_launch"] = fetch * 1024 * 2      # gfx950: FETCH_SIZE = 1/2 of the bytes
            e["hbm_write_bytes_per_launch"] = write * 1024
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
    out["kernels"][k] = e
for extra in ("bench.log", "phase.log"):
    p = os.path.join(ROOT, "gpurun_out", extra)
    if os.path.exists(p):
        lines = [l.rstrip("\n") for l in open(p) if l.startswith("{") or l.startswith("kernel ")]
        if lines:
            out[extra] = lines[-1]
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
# what bench.py reports as roofline.traffic: HBM bytes per launch of the kernels that run the worker loop (the fused
# kernel, or the per-read kernel + the Stats kernel of the split / lane plan), PMC-derived
main = [k for k in out["kernels"] if k == "fq_fused_kernel" or k.startswith("fq_scan") or "fq_lane_kernel" in k or k in ("fq_stats_kernel", "fq_stats5_kernel")]
main = [k for k in main if "hbm_bytes_per_launch" in out["kernels"][k]]
if main:
    tot = {f: sum(out["kernels"][k][f] for k in main) for f in ("hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "hbm_bytes_per_launch")}
    # per PAIR, so that bench.py can attach the figure at whatever batch size it runs (the driver's --steps 20 picks 5.03 M pairs)
    json.dump({"tag": tag, "kernel": " + ".join(main), "pairs_per_launch": PAIRS, **tot,
               "hbm_bytes_per_pair": tot["hbm_bytes_per_launch"] / PAIRS,
               "hbm_read_bytes_per_pair": tot["hbm_read_bytes_per_launch"] / PAIRS,
               "hbm_write_bytes_per_pair": tot["hbm_write_bytes_per_launch"] / PAIRS,
               "per_kernel": {k: dict({f: out["kernels"][k][f] for f in tot}, hbm_bytes_per_pair=out["kernels"][k]["hbm_bytes_per_launch"] / PAIRS) for k in main},
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE x2 (gfx950 halves wide reads)"},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
# VALU / LDS wave-instructions per pair of the same kernels from the SQ passes (gpurun_out/prof/<tag>_sq*)
import glob
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(src, tag + "_sq*"))):
    for fcsv in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fcsv)):
            kn = r["Kernel_Name"]
            if "fq_lane_kernel" in kn or kn in ("fq_stats_kernel", "fq_stats5_kernel", "fq_fused_kernel", "fq_scan_kernel"):
                sq["fq_lane_kernel" if "fq_lane_kernel" in kn else kn][r["Counter_Name"]].append(float(r["Counter_Value"]))
if sq:
    per = {}
    for kn, ctrs in sq.items():
        per[kn] = {c: sum(v) / len(v) for c, v in ctrs.items()}
    valu = sum(p.get("SQ_INSTS_VALU", 0.0) for p in per.values())
    lds = sum(p.get("SQ_INSTS_LDS", 0.0) for p in per.values())
    json.dump({"tag": tag, "kernel": " + ".join(sorted(per)), "pairs_per_launch": PAIRS,
               "insts_valu_per_launch": valu, "insts_valu_per_pair": valu / PAIRS, "insts_lds_per_pair": lds / PAIRS,
               "per_kernel": {kn: {"insts_valu_per_pair": p.get("SQ_INSTS_VALU", 0.0) / PAIRS, "insts_lds_per_pair": p.get("SQ_INSTS_LDS", 0.0) / PAIRS,
                                   "wait_any_frac": (p.get("SQ_WAIT_ANY", 0.0) / p["SQ_WAVE_CYCLES"]) if p.get("SQ_WAVE_CYCLES") else None,
                                   "lds_bank_conflict_frac": (p.get("SQ_LDS_BANK_CONFLICT", 0.0) / p["SQ_LDS_IDX_ACTIVE"]) if p.get("SQ_LDS_IDX_ACTIVE") else None}
                              for kn, p in per.items()},
               "peak_T_lane_ops_per_s": 36.3,
               "method": "SQ_INSTS_VALU of the plan's kernels from rocprofv3 --pmc (profiles/<tag>_sq_counters.txt), one launch of pairs_per_launch pairs; "
                         "'peak' = the measured issue rate of the bit-op / v_bcnt / v_sad_u8 / v_alignbit class (profiles/r02c_issue_rate_microbench.txt: ~4.5 "
                         "cycles per wave64 instruction per SIMD, 1024 SIMDs, 2.4 GHz, x64 lanes); v_add / v_and issue in 2.8 cycles, so the fraction is an upper bound"},
              open(os.path.join(dst, "valu.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
