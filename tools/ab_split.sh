# Parity tests of the Spectrum/LPF kernels, then rates of the split kernel: the tree's build and the
# builds under build_ab/ (tools, not product).
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "split or spectrum or lpf" --timeout 120 --timeout-method thread > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -3 gpurun_out/ab_tests.log
for v in "" $AB_LIBS; do
  echo "== ${v:-tree}"
  BEATDET_LIB=${v:+$PWD/$v} STREAMS=16384 LENGTHS=${AB_LENGTHS:-450,500,517,529,532,540,600} timeout -k 10 120 python tools/exp_split.py
done
