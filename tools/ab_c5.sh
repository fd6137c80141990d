#!/bin/bash
# Whole-step A/B of library builds at config-5 sizes (512 rows) and at the headline sizes, joint phase, production build
# first and last:  AB_LIBS="ab_libs/libTOUCH.so ab_libs/libS5.so" bash tools/ab_c5.sh  [AB_PARITY=1: also the kernel-level
# and single-minibatch parity tests against each variant]
for v in "" ${AB_LIBS} ""; do
  if [ -n "$AB_PARITY" ] && [ -n "$v" ]; then
    PVAE_LIB_PATH=$PWD/$v python -m pytest tests/test_gpu_parity.py tests/test_gpu_shapes.py -x -q -k "gemm or single_batch or odd_minibatch or tiles" 2>&1 | tail -1
  fi
  for cfg in "--config c5" ""; do
    r=$(PVAE_LIB_PATH=${v:+$PWD/$v} python bench.py --inner --phase joint $cfg --steps 400 --warmup 40 2>/dev/null | grep '^{' | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us' % (d['ms_per_step']*1e3))")
    echo "${v:-production} ${cfg:-c2}: $r"
  done
done
