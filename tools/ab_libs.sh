#!/bin/bash
# A/B whole-step timing of library variants: bash tools/ab_libs.sh ab_libs/libA.so ab_libs/libB.so ...
# ("" = the production library).  Variants are built into ab_libs/ (snapshotted by gpurun).
# AB_TEST=1 first runs the kernel-level and single-minibatch parity tests against each variant.
for v in "" "$@" ""; do
  echo "== ${v:-production}"
  if [ -n "$AB_TEST" ]; then
    PVAE_LIB_PATH=${v:+$PWD/$v} python -m pytest tests/test_gpu_parity.py -x -q -k "gemm or single_batch or odd_minibatch or fused_adam" 2>&1 | tail -2
  fi
  PVAE_LIB_PATH=${v:+$PWD/$v} python bench.py --no-cpu-baseline --no-rocprof --steps 400 --warmup 40 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
def row(k):
    return '  '.join('%s %.2f' % (n.split(' (')[0][:18], v['avg_us']) for n, v in k.items() if isinstance(v, dict))
print('joint %.2f us | %s' % (d['ms_per_step']*1e3, row(d['kernels'])))
print('world %.2f us | %s' % (d['world_ms_per_step']*1e3, row(d['world_kernels'])))"
done
