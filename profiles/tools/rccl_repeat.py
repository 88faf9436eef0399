#!/usr/bin/env python3
"""Repeat the world-size-1 RCCL check (tests/dist_device_check.py) N times in fresh processes, alternating libptnn's single-node
defaults (bootstrap over loopback) with RCCL's own choice (PTNN_COMM_KEEP_ENV=1: the pod's interface), and record the wall time
and the stage stamps of every run -- every stage is bounded (PTNN_COMM_TIMEOUT_S), so a stall ends as error -7 naming its stage.
The round-2 record holds ONE silent 300 s stall of this check in some twenty suite runs; this tool looks for a second one.
    python profiles/tools/rccl_repeat.py N out.jsonl"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n, out = int(sys.argv[1]), sys.argv[2]
for k in range(n):
    keep = k % 2
    env = dict(os.environ, PTNN_COMM_TRACE="1", PTNN_COMM_TIMEOUT_S="45")
    if keep:
        env["PTNN_COMM_KEEP_ENV"] = "1"
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_device_check.py")], capture_output=True, text=True, timeout=200, env=env)
        rc, err, so = r.returncode, r.stderr, r.stdout
    except subprocess.TimeoutExpired as e:
        rc, err, so = -999, (e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or ""), ""
    dt = time.time() - t0
    stamps = [l for l in err.splitlines() if l.startswith("[ptnn comm") and ("dlopen" in l or "ncclCommInitRank" in l or "ncclGetUniqueId" in l)][:8]
    rec = dict(run=k, keep_env=keep, rc=rc, seconds=round(dt, 2), ok=("OK no torch" in so), first_stamps=stamps, tail=err.splitlines()[-3:] if rc else [])
    with open(out, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print(json.dumps({k_: rec[k_] for k_ in ("run", "keep_env", "rc", "seconds", "ok")}), flush=True)
