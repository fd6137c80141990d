for v in "" ab_libs/libpvae_bsum1.so ab_libs/libpvae_bsum2.so ""; do
  echo "== ${v:-production}"
  PVAE_LIB_PATH=${v:+$PWD/$v} python bench.py --no-cpu-baseline --steps 600 --warmup 60 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('world %.2f us  joint %.2f us  bwd_pair %.2f us  wgrad %.2f us' % (d['ms_per_step']*1e3, d['joint_ms_per_step']*1e3, d['roofline']['avg_launch_us'], [v for k, v in d['kernels'].items() if 'wgrad_reg' in k][0]['avg_us']))"
done
