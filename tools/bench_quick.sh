#!/bin/bash
# joint / world step time and per-category launch averages of one library: bash tools/bench_quick.sh [bench flags]
python bench.py --no-cpu-baseline --no-rocprof --steps 400 --warmup 40 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
def row(k):
    return '  '.join('%s %.2f' % (n.split(' (')[0][:18], v['avg_us']) for n, v in k.items() if isinstance(v, dict))
print('joint %.2f us | %s' % (d['ms_per_step']*1e3, row(d['kernels'])))
print('world %.2f us | %s' % (d['world_ms_per_step']*1e3, row(d['world_kernels'])))"
