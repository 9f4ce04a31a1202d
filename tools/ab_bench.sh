# Headline bench, alternating between the tree's build and another build of the library
# (tools, not product).  Usage: ab_bench.sh <other .so> [bench args]
set -e
OTHER=$1; shift
for r in 1 2 3; do
  for v in "" "$OTHER"; do
    BEATDET_LIB=${v:+$PWD/$v} timeout -k 10 200 python bench.py --no-cpu-baseline --also none --steps 10 --warmup 3 "$@" > /tmp/ab.json
    python - "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1] or 'tree':24s} ms_per_step {d['ms_per_step']:.3f} kernel_ms {d['roofline']['kernel_ms']:.3f} achieved {d['roofline']['achieved']:.0f}")
PY
  done
done
