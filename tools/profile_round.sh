#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   bash tools/profile_round.sh r01
# One kernel-trace + stats pass per phase, then one --pmc pass per counter group (counters are
# never combined with the hip/hsa trace domains).  Output goes to gpurun_out/<tag>*; condense with
#   python tools/summarize_profiles.py gpurun_out/<tag> profiles/<tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/${TAG}"_* "$O/${TAG}j_"*
CMD="python $ROOT/bench.py --steps 100 --warmup 20 --no-extra --no-cpu-baseline"
CMDJ="$CMD --phase joint"

rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_trace" -o t -- $CMD \
    > "$O/${TAG}_trace.json" 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}j_trace" -o t -- $CMDJ \
    > "$O/${TAG}j_trace.json" 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/${TAG}_fetch" -o f -- $CMD \
    > /dev/null 2> "$O/${TAG}_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/${TAG}_write" -o w -- $CMD \
    > /dev/null 2> "$O/${TAG}_write.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace \
    --output-format csv -d "$O/${TAG}_mfma" -o m -- $CMD > /dev/null 2> "$O/${TAG}_mfma.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace \
    --output-format csv -d "$O/${TAG}_lds" -o l -- $CMD > /dev/null 2> "$O/${TAG}_lds.err"

# plain bench lines (no profiler attached) for the committed JSON
python "$ROOT/bench.py" > "$O/bench_default.json" 2> "$O/bench_default.err"
python "$ROOT/bench.py" --config c5 --no-cpu-baseline > "$O/bench_c5.json" 2> "$O/bench_c5.err"
find "$O" -name '*.db' -delete
ls "$O/${TAG}"_* | head -40
tail -2 "$O/${TAG}_fetch.err"
cat "$O/bench_default.json"
