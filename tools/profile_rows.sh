#!/bin/bash
# rocprofv3 evidence for the joint step at a given number of rows per GPU (run through gpurun):
#   bash tools/profile_rows.sh r04 4096      ->  gpurun_out/<tag>_b<rows>_{trace,fetch,write,mfma,lds}
# condense with  python tools/summarize_profiles.py gpurun_out/<tag>_b<rows> profiles/<tag>_b<rows>
set -u
TAG=${1:-r04}; ROWS=${2:-4096}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --inner --steps 40 --warmup 10 --phase joint --batch $ROWS"
P="$O/${TAG}_b${ROWS}"
rm -rf "${P}"_trace "${P}"_fetch "${P}"_write "${P}"_mfma "${P}"_lds
rocprofv3 --kernel-trace --stats --output-format csv -d "${P}_trace" -o t -- $CMD > /dev/null 2> "${P}_trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "${P}_fetch" -o f -- $CMD > /dev/null 2> "${P}_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "${P}_write" -o w -- $CMD > /dev/null 2> "${P}_write.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "${P}_mfma" -o m -- $CMD > /dev/null 2> "${P}_mfma.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d "${P}_lds" -o l -- $CMD > /dev/null 2> "${P}_lds.err"
python $ROOT/tools/step_trace.py "${P}_trace" > "${P}_step.txt" 2>&1
find "$O" -name '*.db' -delete
python $ROOT/tools/summarize_profiles.py "${P}" "${P}_summary" 2>&1 | tail -3
