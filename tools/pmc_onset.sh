#!/bin/bash
# SQ counters of the onset kernels (one pass, 8 SQ counters). Output: gpurun_out/prof/onset_sq
set -u
OUT=gpurun_out/prof/onset_sq
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT -- \
    python3 tools/bench_onset.py --streams 1024 --reps 2 "$@" > $OUT/bench.json 2> $OUT/err.txt || exit 1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/prof/onset_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        a = acc[row["Kernel_Name"][:40]][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    if "bd_" in k:
        print(k, {c: round(v[0] / v[1]) for c, v in sorted(cs.items())})
PY
cat $OUT/bench.json
