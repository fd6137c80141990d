#!/bin/bash
# A/B of runtime switches on the un-profiled step (joint and world): bash tools/ab_env2.sh "PVAE_DGRAD16=0" "PVAE_KROT=1" ...
for v in "" "$@" ""; do
  for ph in joint world; do
    r=$(env $v python bench.py --inner --phase $ph --steps 400 --warmup 40 2>/dev/null | grep '^{' | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us' % (d['ms_per_step']*1e3))")
    echo "${v:-defaults} $ph: $r"
  done
done
