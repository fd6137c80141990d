#!/bin/bash
# per-kernel durations of the timed phase: bash tools/kernel_trace.sh <tag> [bench flags]   -> gpurun_out/<tag>_kernels.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o t -- python $ROOT/bench.py --inner --steps 100 --warmup 20 "$@" > /dev/null 2>&1
python - "$TAG" <<'PY' > $O/${TAG}_kernels.txt
import csv, glob, sys
f = glob.glob("/tmp/kt_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if not r["Name"].startswith(("void at::", "__amd", "at::"))]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.1f us" % (tot / 100 / 1e3 * 100 / 120))
for r in rows[:24]:
    print("%8.2f us x %5d  %5.1f%%  %s" % (float(r["AverageNs"]) / 1e3, int(r["Calls"]), 100 * float(r["TotalDurationNs"]) / tot, r["Name"][:150]))
PY
cat $O/${TAG}_kernels.txt
