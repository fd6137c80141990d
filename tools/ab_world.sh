#!/bin/bash
# whole-step A/B of library builds, world phase (configs[1]) and joint phase: AB_LIBS="ab_libs/x.so ..." bash tools/ab_world.sh
for v in "" $AB_LIBS ""; do
  for ph in world joint; do
    r=$(PVAE_LIB_PATH=${v:+$PWD/$v} python bench.py --inner --phase $ph --steps 400 --warmup 40 2>/dev/null | grep '^{' | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us' % (d['ms_per_step']*1e3))")
    echo "${v:-production} $ph: $r"
  done
done
