#!/usr/bin/env python3
"""Condenses a tools/profile_bench.sh output directory into a small text
summary (kernel stats + HBM traffic per launch of each kernel) for profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
lines = []
for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append(f"== rocprofv3 --kernel-trace --stats ({os.path.basename(f)}) ==")
    with open(f) as fh:
        for row in csv.DictReader(fh):
            lines.append(f"{row['Name'][:70]:70s} calls={row['Calls']:>4s} avg_ns={row['AverageNs']:>14s} "
                         f"total_ns={row['TotalDurationNs']:>14s} pct={row['Percentage']}")
traffic = {}
for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(src, tag, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [0.0, 0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = row["Kernel_Name"][:70]
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
        lines.append(f"== rocprofv3 --pmc {counter} (per-launch average, raw counter units = KiB) ==")
        for k, (v, n) in acc.items():
            lines.append(f"{k:70s} launches={n:4d} {counter}_avg={v / n:.1f}")
            traffic.setdefault(k, {})[counter] = v / n
for f in glob.glob(os.path.join(src, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"][:70]
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    lines.append("== rocprofv3 --pmc SQ_* (per-launch averages; SQ cycle counters are in quad-cycles summed over waves/SEs) ==")
    for k, cs in acc.items():
        if "bd_" not in k:
            continue
        v = {c: x[0] / x[1] for c, x in cs.items()}
        lines.append(f"{k:70s} " + " ".join(f"{c}={val:.4g}" for c, val in sorted(v.items())))
        if v.get("SQ_LDS_IDX_ACTIVE"):
            lines.append(f"{'':70s} lds_bank_conflict_share={v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE']:.4f}")
        if v.get("SQ_WAVE_CYCLES"):
            lines.append(f"{'':70s} valu_active_share_of_wave_cycles={v.get('SQ_ACTIVE_INST_VALU', 0) / v['SQ_WAVE_CYCLES']:.4f} "
                         f"wait_any_share={v.get('SQ_WAIT_ANY', 0) / v['SQ_WAVE_CYCLES']:.4f}")
        if v.get("SQ_BUSY_CYCLES"):
            lines.append(f"{'':70s} valu_active_per_busy_cycle={v.get('SQ_ACTIVE_INST_VALU', 0) / v['SQ_BUSY_CYCLES']:.4f}")
for j in ("bench_trace.json", "bench_fetch.json", "bench_write.json"):
    p = os.path.join(src, j)
    if os.path.exists(p):
        lines.append(f"== {j} ==")
        lines.append(open(p).read().strip())
lines.append("== HBM traffic per launch, corrected for gfx950 (FETCH_SIZE counts 64 B per 128-B request on wide "
             "coalesced streaming reads: x2; WRITE_SIZE exact); bytes = KiB * 1024 ==")
for k, t in traffic.items():
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        b = (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024
        lines.append(f"{k:70s} hbm_bytes_per_launch={b:.0f}")
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
