#!/bin/bash
# A/B of runtime switches: bash tools/ab_env.sh "PVAE_WGRAD32=0" "PVAE_KROT=1" ...  ("" = defaults)
for v in "" "$@" ""; do
  echo "== ${v:-defaults}"
  env $v python bench.py --no-cpu-baseline --steps 600 --warmup 60 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernels']
print('world %.2f us  joint %.2f us | ' % (d['ms_per_step']*1e3, d['joint_ms_per_step']*1e3) + '  '.join('%s %.2f' % (n.split(' ')[0][:22], v['avg_us']) for n, v in k.items()))"
done
