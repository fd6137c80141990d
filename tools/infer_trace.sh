#!/bin/bash
# Kernel trace of the rollout forward at B = 1 (default sizes): per-launch durations and the gap between dependent
# launches, averaged over the last 50 calls.  bash tools/infer_trace.sh   (GPU box, through gpurun)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/infer_trace_driver.py <<PY
import sys, os, contextlib, io, tempfile, torch
sys.path.insert(0, "$ROOT"); sys.path.insert(0, os.path.join("$ROOT", "tools"))
from synth_demo import synth_demo, write_demo
from physicsvae_amd import train_physics_vae as T
td = tempfile.mkdtemp(); write_demo(os.path.join(td, "d.pkl"), synth_demo(0, 2, 50, 197, 45))
T.args = T.arg_parser().parse_args(["--data_train", os.path.join(td, "d.pkl"), "--batch_size", "32"])
cfg = T.get_trainer_config(T.args); cfg["model"]["custom_model_config"]["device"] = "cuda"
with contextlib.redirect_stdout(io.StringIO()):
    tr = T.TrainModel(cfg)
obs = torch.randn(1, 394, device="cuda"); out = None
for _ in range(300):
    out = tr.engine.infer(obs, want_s2=False, out=out)
torch.cuda.synchronize()
PY
rm -rf /tmp/infer_trace_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/infer_trace_out -o t -- python /tmp/infer_trace_driver.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/infer_trace_out/**/*kernel_trace.csv", recursive=True)[0])))
rows = sorted((r for r in rows if "gemv_rollout" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 300
last = rows[-n * 50:]
per = collections.defaultdict(list)
for i, r in enumerate(last):
    per[i % n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
gaps = sorted((int(last[i]["Start_Timestamp"]) - int(last[i - 1]["End_Timestamp"])) / 1e3 for i in range(1, len(last)))
print("launches per call", n, "| durations us", [round(sum(v) / len(v), 2) for _, v in sorted(per.items())],
      "| workgroups", [int(last[i]["Grid_Size_X"]) // 256 for i in range(n)], "| median gap us", gaps[len(gaps) // 2])
PY
