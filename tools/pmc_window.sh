#!/bin/bash
# Counters of the Spectrum window kernel at one callback length (two passes: SQ, then TCP/TA).
# Usage: pmc_window.sh <callback_len>
set -u
L=${1:-441}
OUT=gpurun_out/prof/win_$L
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cat > $OUT/run.py <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import beat_detector_amd as bd
sr = 44100; streams = 2048; n = sr * 60; stride = (n + 63) & ~63
buf = torch.empty((streams, stride), dtype=torch.int16, device="cuda")
bd.synth_fill(buf, n_samples=n, seed=1)
for _ in range(3):
    bd.batch_candidates(bd.StrategyKind.Spectrum, sr, buf, $L, n_samples=n)
torch.cuda.synchronize()
PY
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python3 $OUT/run.py > $OUT/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum --output-format csv -d $OUT/tcp -- python3 $OUT/run.py > $OUT/tcp.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $OUT/run.py > $OUT/trace.log 2>&1 || exit 3
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        a = acc[row["Kernel_Name"][:40]][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    if "spectrum" in k:
        print(k, {c: round(v[0] / v[1]) for c, v in sorted(cs.items())})
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "spectrum" in row["Name"]:
            print(row["Name"][:50], row["Calls"], row["AverageNs"])
PY
