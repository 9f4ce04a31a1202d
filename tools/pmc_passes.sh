#!/bin/bash
# Hardware-counter passes over bench.py for one workload, one rocprofv3 run per pass (a pass holds at most
# 8 SQ, 4 TCP, 4 TCC, 2 TA, 2 GRBM counters).  Usage: pmc_passes.sh <tag> <kernel substring> [bench args]
# Prints per-launch averages of every counter for kernels whose name contains the substring.
set -u
TAG=$1; KERNEL=$2; shift 2
OUT=gpurun_out/prof/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
PASSES=(
 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_WR"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum"
 "TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WR_UNCACHED_32B_sum"
 "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_WRITE_sum"
 "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_NC_WRITE_REQ_sum TCP_TCC_UC_WRITE_REQ_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
)
# PMC_ONLY="<counters>" runs that one pass instead
if [ -n "${PMC_ONLY:-}" ]; then PASSES=("$PMC_ONLY"); fi
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -- \
     python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --also none "$@" > $OUT/bench_p$i.json 2> $OUT/p$i.err || { echo "pass $i failed"; tail -3 $OUT/p$i.err; exit 1; }
done
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, sys, json
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
res = {c: v[0] / v[1] for c, v in sorted(acc.items())}
for c, v in res.items():
    print(f"{c:45s} {v:.4g}")
ms = [json.loads(open(f).read().strip().splitlines()[-1])["roofline"]["kernel_ms"] for f in glob.glob(sys.argv[1] + "/bench_p*.json")]
print("kernel_ms under the counter passes:", [round(m, 3) for m in ms])
PY
