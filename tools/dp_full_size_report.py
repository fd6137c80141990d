"""Run tests/dp_full_size_worker.py for several exchange forms (8 ranks, sharing GPUs when the box has fewer) and print what
each run measured: `python tools/dp_full_size_report.py c3 p2p p2p_push default p2p` (a form named twice runs twice)."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
config, forms = sys.argv[1], sys.argv[2:]
world = int(os.environ.get("DP_REPORT_RANKS", "8"))
ndev = max(torch.cuda.device_count(), 1)
for i, form in enumerate(forms):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = os.path.join(tempfile.mkdtemp(), "r")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0", PVAE_DP_FORMS=form)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_full_size_worker.py"), ROOT, out, config],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r % ndev), PVAE_LOCAL_DEVICE=str(r % ndev)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    if any(p.returncode for p in procs):
        print(form, "FAILED", outs[0][-1500:])
        continue
    res = torch.load(out + ".0")[form]
    for ph in ("world", "joint"):
        print(json.dumps({"config": config, "form": form, "run": i, "phase": ph, **res[ph]}), flush=True)
