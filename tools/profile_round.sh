#!/bin/bash
# Runs on the GPU box: tools/profile_bench.sh for every bench workload of the round, condensed into
# profiles/<round>_<workload>.txt (+ the kernel-stats CSV).  Usage: profile_round.sh <round tag> [workloads...]
set -u
ROUND=${1:-r2}; shift || true
# A workload name may carry the suffix _decisions: the launch that writes decisions only (--outputs decisions);
# without it the two strategies are profiled with detection values written (--outputs values).
WORKLOADS=${*:-spectrum_hop512 spectrum_hop512_decisions spectrum_1024 spectrum_529 spectrum_441 lpf biquad onset_2048 onset_2048_power}
mkdir -p gpurun_out/profiles
for W in $WORKLOADS; do
  case $W in
    *_decisions) EXTRA="--workload ${W%_decisions} --outputs decisions" ;;
    spectrum_*|lpf*) EXTRA="--workload $W --outputs values" ;;
    *) EXTRA="--workload $W" ;;
  esac
  bash tools/profile_bench.sh ${ROUND}_$W $EXTRA > gpurun_out/prof_${ROUND}_$W.log 2>&1 || { echo "profile of $W failed"; tail -5 gpurun_out/prof_${ROUND}_$W.log; exit 1; }
  python3 tools/summarize_prof.py gpurun_out/prof/${ROUND}_$W gpurun_out/profiles/${ROUND}_$W.txt > /dev/null
  cp $(find gpurun_out/prof/${ROUND}_$W/trace -name "*kernel_stats.csv" | head -1) gpurun_out/profiles/${ROUND}_${W}_kernel_stats.csv
  echo "profiled $W"
done
