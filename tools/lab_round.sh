#!/bin/bash
# One-off measurement batch (run through gpurun); writes gpurun_out/lab_*.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p "$O"
cd "$ROOT"
AB_TEST=1 bash tools/ab_libs.sh ab_libs/lib_base.so ab_libs/lib_sc1.so 2>&1 | tee "$O/lab_ab.log"
echo "=== k64 failure"
PVAE_LIB_PATH=$ROOT/ab_libs/lib_k64.so python -m pytest tests/test_gpu_parity.py -x -q -k "gemm_wgrad" 2>&1 | tail -30 | tee "$O/lab_k64.log"
cd /tmp && export TMPDIR=/tmp
for v in base sc1; do
  rm -rf "$O/lab_trace_$v"
  PVAE_LIB_PATH=$ROOT/ab_libs/lib_$v.so rocprofv3 --kernel-trace --output-format csv -d "$O/lab_trace_$v" -o t -- \
      python $ROOT/bench.py --inner --phase world --steps 100 --warmup 20 > /dev/null 2>&1
  echo "=== step trace $v"
  python $ROOT/tools/step_trace.py "$(dirname $(find $O/lab_trace_$v -name '*kernel_trace.csv' | head -1))" | tee "$O/lab_step_$v.txt"
  find "$O/lab_trace_$v" -name '*.db' -delete
done
rocprofv3 -L > "$O/lab_counters.txt" 2>&1
grep -c . "$O/lab_counters.txt"
rm -rf "$O/lab_pmcA"
PVAE_LIB_PATH=$ROOT/ab_libs/lib_base.so rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS \
    --kernel-trace --output-format csv -d "$O/lab_pmcA" -o p -- python $ROOT/bench.py --inner --phase world --steps 60 --warmup 10 > /dev/null 2> "$O/lab_pmcA.err"
python - <<PY
import csv, glob, collections
fs = glob.glob("$O/lab_pmcA/**/*counter_collection.csv", recursive=True)
if fs:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        k = "bwd_pair" if "bwd_pair" in n else "fwd_ws" if "splitk_ws_kernel<true" in n else "wgrad_pair" if "wgrad_pair" in n else "reg16" if "reg16" in n else None
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
else:
    print(open("$O/lab_pmcA.err").read()[-1500:])
PY
find "$O/lab_pmcA" -name '*.db' -delete
