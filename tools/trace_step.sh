#!/bin/bash
# One kernel-trace pass of the joint (or given) step and its per-launch listing: bash tools/trace_step.sh <tag> [bench flags]
set -u
TAG=${1:-t}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/${TAG}_trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_trace" -o t -- python $ROOT/bench.py --inner --steps 100 --warmup 20 "$@" > /dev/null 2> "$O/${TAG}_trace.err"
python $ROOT/tools/step_trace.py "$O/${TAG}_trace" > "$O/${TAG}_step.txt" 2>&1
find "$O/${TAG}_trace" -name '*.db' -delete
cat "$O/${TAG}_step.txt"
