#!/usr/bin/env python3
"""Per-step parity at the BASELINE run lengths (tests/parity.py: follow_device_run; the suite runs shorter chains to stay within
minutes): every MH step, cascade decision and trace row of whole runs of the bench's own workloads against the C oracle, with
the device's decisions AND state imposed (per-step errors; accepted steps re-evaluated at the device's own proposal), and the
drift form beside it (decisions only).
    python profiles/tools/follow_full.py out.jsonl      # on the GPU box; a few minutes
One JSON report per workload: steps, decisions the oracle would have taken differently, largest log-alpha error by class."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = sys.argv[1] if len(sys.argv) > 1 else "/dev/null"
os.environ.setdefault("PTNN_PARITY_PROBE", "/dev/null")      # measure, do not judge the bounds
os.environ["PTNN_FOLLOW_DRIFT"] = "1"

import numpy as np  # noqa: E402
import test_gpu_follow as tf  # noqa: E402
import ptnn_amd  # noqa: E402,F401
from ptnn_amd import philox  # noqa: E402

# name: task, topology, data set, replicas, Langevin, lr, maxtemp, samples per replica, swap interval (bench.py WORKLOADS), noise modes
FULL = {
    "sunspot64": (0, (4, 5, 1), "sunspot", 64, True, 0.1, 2, 10000, 100, (1, 0)),
    "iris16": (1, (4, 12, 3), "iris", 16, False, 0.01, 10, 10000, 100, (1,)),
    "mackey64": (0, (4, 10, 1), "mackey", 64, True, 0.1, 2, 10000, 100, (1,)),
    "ionosphere256": (1, (34, 50, 2), "ions", 256, False, 0.01, 10, 3000, 100, (1,)),
    "sunspot5_r4": (0, (5, 5, 1), "sunspot5", 4, True, 0.1, 2, 10000, 100, (1,)),
    "mackey5_r64": (0, (5, 10, 1), "mackey5", 64, True, 0.1, 2, 10000, 100, (1,)),
}
ONLY = os.environ.get("FOLLOW_ONLY")                       # e.g. FOLLOW_ONLY=ionosphere256
for name, (task, topo, dname, R, lg, lr, mt, S, si, modes) in FULL.items():
    if ONLY and name not in ONLY.split(","):
        continue
    for sn in modes:
        t0 = time.time()
        w0 = np.stack([philox.initial_weights(1, r, topo[0] * topo[1] + topo[1] * topo[2] + topo[1] + topo[2]) for r in range(R)])
        rep = tf.followed_run(task, topo, dname, R, lg, lr, mt, S, si, 1, f"{name} full shared_noise={sn} ", shared_noise=sn, w0=w0)
        rep.update(workload=name, R=R, S=S, shared_noise=sn, seconds=round(time.time() - t0, 1))
        with open(OUT, "a") as f:
            f.write(json.dumps(rep) + "\n")
        print(json.dumps(rep), flush=True)
