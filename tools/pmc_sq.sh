#!/bin/bash
# One pass of 8 SQ counters over bench.py for a workload. Usage: pmc_sq.sh <workload> [extra bench args]
set -u
W=$1; shift
OUT=gpurun_out/prof/sq_$W
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT -- \
    python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.txt || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        a = acc[row["Kernel_Name"][:40]][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    if "bd_" in k and "synth" not in k:
        print(k, {c: round(v[0] / v[1]) for c, v in sorted(cs.items())})
PY
grep -o '"kernel_ms": [0-9.]*' $OUT/bench.json
