#!/bin/bash
# A/B whole-step timing of library variants: bash tools/ab_libs.sh ab_libs/libA.so ab_libs/libB.so ...
# ("" = the production library).  Variants are built into ab_libs/ (snapshotted by gpurun).
for v in "" "$@" ""; do
  echo "== ${v:-production}"
  PVAE_LIB_PATH=${v:+$PWD/$v} python bench.py --no-cpu-baseline --steps 600 --warmup 60 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('world %.2f us  joint %.2f us | ' % (d['ms_per_step']*1e3, d['joint_ms_per_step']*1e3) + '  '.join('%s %.2f' % (n.split(' ')[0][:22], v['avg_us']) for n, v in k.items()))"
done
