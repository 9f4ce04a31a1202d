#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace + stats of bench.py, then the HBM
# traffic counters in separate passes (FETCH_SIZE and WRITE_SIZE do not fit in
# one pass).  Output under gpurun_out/prof/<tag>/.
set -u
TAG=${1:-r1}
shift || true
EXTRA="$*"
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- \
    python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --also none $EXTRA > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --also none $EXTRA > $OUT/bench_fetch.json 2> $OUT/fetch.err || exit 2
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --also none $EXTRA > $OUT/bench_write.json 2> $OUT/write.err || exit 3
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --also none $EXTRA > $OUT/bench_sq.json 2> $OUT/sq.err || exit 4
find $OUT -name "*.csv" | head -20
