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
