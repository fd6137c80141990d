#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   bash tools/profile_round.sh r02
# Per workload (joint = the headline step, world = BASELINE configs[1], c5 = configs[4] sizes, joint): one
# kernel-trace + stats pass, then one --pmc pass per counter group (counters are never combined with the
# hip/hsa trace domains).  Output goes to gpurun_out/<tag>*; condense with
#   python tools/summarize_profiles.py gpurun_out/<tag>_<workload> profiles/<tag>_<workload>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BASE="python $ROOT/bench.py --inner --steps 100 --warmup 20"
for W in joint world c5; do
  case $W in
    joint) CMD="$BASE --phase joint" ;;
    world) CMD="$BASE --phase world" ;;
    c5)    CMD="$BASE --phase joint --config c5" ;;
  esac
  P="$O/${TAG}_${W}"
  rm -rf "${P}"_trace "${P}"_fetch "${P}"_write "${P}"_mfma "${P}"_lds
  rocprofv3 --kernel-trace --stats --output-format csv -d "${P}_trace" -o t -- $CMD > /dev/null 2> "${P}_trace.err"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "${P}_fetch" -o f -- $CMD > /dev/null 2> "${P}_fetch.err"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "${P}_write" -o w -- $CMD > /dev/null 2> "${P}_write.err"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace \
      --output-format csv -d "${P}_mfma" -o m -- $CMD > /dev/null 2> "${P}_mfma.err"
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace \
      --output-format csv -d "${P}_lds" -o l -- $CMD > /dev/null 2> "${P}_lds.err"
  python $ROOT/tools/step_trace.py "${P}_trace" > "${P}_step.txt" 2>&1
done
# plain bench lines (no profiler attached around them; the default line runs its own rocprofv3 children)
cd "$ROOT"
python bench.py > "$O/${TAG}_bench_default.json" 2> "$O/${TAG}_bench_default.err"
python bench.py --phase world --no-cpu-baseline > "$O/${TAG}_bench_world.json" 2> "$O/${TAG}_bench_world.err"
python bench.py --config c5 --no-cpu-baseline > "$O/${TAG}_bench_c5.json" 2> "$O/${TAG}_bench_c5.err"
python bench.py --gpus 2 --no-cpu-baseline | grep "^{" > "$O/${TAG}_bench_n2_shared_gpu.json" 2> "$O/${TAG}_bench_n2.err"
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -12 > "$O/${TAG}_hw.txt"; lscpu | head -20 >> "$O/${TAG}_hw.txt"
find "$O" -name '*.db' -delete
ls "$O" | grep "^${TAG}_" | head -60
tail -c 600 "$O/${TAG}_bench_default.json"
