for v in "" ${AB_LIBS:-ab_libs/libD1.so ab_libs/libD3.so ab_libs/libW1.so ab_libs/libW3.so} ""; do
  for cfg in "--config c5" ""; do
    r=$(PVAE_LIB_PATH=${v:+$PWD/$v} python bench.py --inner --phase joint $cfg --steps 400 --warmup 40 2>/dev/null | grep '^{' | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f us' % (d['ms_per_step']*1e3))")
    echo "${v:-production} ${cfg:-c2}: $r"
  done
done
