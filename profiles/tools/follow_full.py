#!/usr/bin/env python3
"""Follow-mode parity at the BASELINE run lengths (tests/parity.py: follow_device_run; the suite runs shorter chains to stay
within minutes): every MH step, cascade decision and trace row of whole runs of the bench's own workloads against the C oracle.
    python profiles/tools/follow_full.py [out.jsonl]      # on the GPU box; a few minutes
Prints one JSON report per workload: steps, decisions the oracle would have taken differently, largest log-alpha error / scale."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PTNN_PARITY_PROBE", sys.argv[1] if len(sys.argv) > 1 else "/dev/null")   # measure, do not judge the bounds

import numpy as np  # noqa: E402
import test_gpu_follow as tf  # noqa: E402

# name: task, topology, data set, replicas, Langevin, lr, maxtemp, samples per replica, swap interval (bench.py WORKLOADS)
FULL = {
    "iris16": (1, (4, 12, 3), "iris", 16, False, 0.01, 10, 10000, 100),
    "mackey64": (0, (4, 10, 1), "mackey", 64, True, 0.1, 2, 10000, 100),
    "ionosphere256": (1, (34, 50, 2), "ions", 256, False, 0.01, 10, 3000, 100),
}
ONLY = os.environ.get("FOLLOW_ONLY")                       # e.g. FOLLOW_ONLY=ionosphere256
for name, (task, topo, dname, R, lg, lr, mt, S, si) in FULL.items():
    if ONLY and name not in ONLY.split(","):
        continue
    t0 = time.time()
    rep = tf.followed_run(task, topo, dname, R, lg, lr, mt, S, si, 1, f"{name} full ", shared_noise=1)
    rep["workload"], rep["S"], rep["seconds"] = name, S, round(time.time() - t0, 1)
    print(json.dumps(rep), flush=True)
