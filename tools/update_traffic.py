#!/usr/bin/env python3
"""Rebuilds profiles/traffic.json (HBM bytes per launch of each workload's dominant kernel) from the
round's profile summaries profiles/<round>_<workload>.txt.  Usage: update_traffic.py <round tag>"""
import glob, json, os, re, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
import bench
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
path = os.path.join(root, "profiles", "traffic.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out["_comment"] = ("HBM bytes per launch of the dominant kernel from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of "
                   "`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --also none --workload W --outputs values` (entries named W_decisions: `--outputs decisions`); bytes = (2*FETCH_SIZE + "
                   "WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 64 B per 128-B request of a wide coalesced read). Source: profiles/<round>_<workload>.txt")
for f in sorted(glob.glob(os.path.join(root, "profiles", tag + "_*.txt"))):
    w = os.path.basename(f)[len(tag) + 1:-4]            # workload, or workload_decisions (the launch without values)
    base = w[:-len("_decisions")] if w.endswith("_decisions") else w
    if base not in bench.WORKLOADS:
        continue
    text = open(f).read()
    section = text.split("== HBM traffic per launch")[-1]
    best = None
    for line in section.splitlines():
        m = re.search(r"^(.*?)\s+hbm_bytes_per_launch=(\d+)", line)
        if m and "bd_" in m.group(1) and "synth" not in m.group(1) and "refractory" not in m.group(1) and "peak_pick" not in m.group(1) \
                and "generic" not in m.group(1):
            if best is None or int(m.group(2)) > best[1]:
                best = (m.group(1).strip(), int(m.group(2)))
    if best:
        out[w] = {"streams": bench.WORKLOADS[base][2], "seconds": bench.WORKLOADS[base][3], "hbm_bytes_per_launch": best[1],
                  "kernel": best[0][:60], "source": "profiles/" + os.path.basename(f)}
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out.items() if isinstance(v, dict)}, indent=1))
