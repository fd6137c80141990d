set -x
mkdir -p gpurun_out/r4e
for b in 128 256 512 1024 2048 4096; do
  python bench.py --batch $b --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r4e/sweep_c2_$b.json
done
for b in 512 1024 2048 4096; do
  python bench.py --config c5 --batch $b --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r4e/sweep_c5_$b.json
done
for b in 1024 4096; do
  python bench.py --phase world --batch $b --no-extra --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r4e/sweep_world_$b.json
done
