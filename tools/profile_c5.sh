set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --inner --steps 100 --warmup 20 --phase joint --config c5"
P="$O/r02_c5"
rm -rf "${P}"_trace "${P}"_fetch "${P}"_write "${P}"_mfma "${P}"_lds
rocprofv3 --kernel-trace --stats --output-format csv -d "${P}_trace" -o t -- $CMD > /dev/null 2> "${P}_trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "${P}_fetch" -o f -- $CMD > /dev/null 2> "${P}_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "${P}_write" -o w -- $CMD > /dev/null 2> "${P}_write.err"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "${P}_mfma" -o m -- $CMD > /dev/null 2> "${P}_mfma.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d "${P}_lds" -o l -- $CMD > /dev/null 2> "${P}_lds.err"
python $ROOT/tools/step_trace.py "${P}_trace" > "${P}_step.txt" 2>&1
cd "$ROOT"
python bench.py --config c5 --no-cpu-baseline > "$O/r02_bench_c5.json" 2> "$O/r02_bench_c5.err"
find "$O" -name '*.db' -delete
tail -c 300 "$O/r02_bench_c5.json"
