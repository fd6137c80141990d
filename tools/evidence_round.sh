#!/bin/bash
# Everything a round's evidence needs from ONE GPU lease (run through gpurun):  bash tools/evidence_round.sh r06
#   rocprofv3 traces + PMC passes + bench lines (tools/profile_round.sh), the 8-rank bench line under the scaling run's
#   launcher (ranks share the GPU on a 1-GPU box), the full GPU test suite, the configs[2] run twice (bit-identical?).
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out; mkdir -p "$O"
cd "$ROOT"
bash tools/profile_round.sh $TAG > "$O/${TAG}_profile_round.log" 2>&1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 8 \
    --steps 200 --warmup 20 > "$O/${TAG}_bench_n8_shared_gpu_torchrun.raw" 2> "$O/${TAG}_bench_n8.err"
grep '^{' "$O/${TAG}_bench_n8_shared_gpu_torchrun.raw" | tail -1 > "$O/${TAG}_bench_n8_shared_gpu_torchrun.json"
# staged (default) against direct first layers (PVAE_DIRECT=1), alternating in this lease (docs/experiments.md: SURVEY K5)
(for i in 1 2; do echo "== staged"; bash tools/bench_quick.sh; echo "== direct"; PVAE_DIRECT=1 bash tools/bench_quick.sh; done) > "$O/${TAG}_ab_direct.txt" 2>&1
python tools/full_run.py --twice > "$O/${TAG}_full_run_twice.txt" 2>&1
(time python -m pytest tests -q -m gpu --durations=15) > "$O/${TAG}_gpu_suite.txt" 2>&1
tail -3 "$O/${TAG}_gpu_suite.txt"; tail -2 "$O/${TAG}_full_run_twice.txt"; tail -c 400 "$O/${TAG}_bench_n8_shared_gpu_torchrun.json"; tail -c 500 "$O/${TAG}_bench_default.json"
