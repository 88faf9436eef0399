#!/usr/bin/env python3
"""Where does a whole run's wall time go outside the segment kernels?  (Ionosphere, 256 replicas: 120 ms per
run against 100 launches x 0.94 ms.)  Times, un-profiled, with host clocks around device synchronisations:
  set_state (restart of the chains) | the run in chunks of `chunk` MH steps, each followed by a sync
for a few consecutive runs, once with HIP-event timing of every launch and once without (PTNN_TIMING_STRIDE=0).
    python profiles/tools/gap_probe.py [workload] [chunk]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(workload, chunk):
    import argparse
    import numpy as np
    import bench
    wl = dict(bench.WORKLOADS[workload])
    train, test, _ = bench.load_data(wl["data"])
    a = argparse.Namespace(waves=0, schedule=0, groups=0, bf16=False, shared_noise=1)
    lad = bench.Ladder(wl, a, train, test, 0, 1, 0)
    s = lad.s
    out = {"stride": os.environ.get("PTNN_TIMING_STRIDE", "1"), "runs": []}
    for run in range(4):
        t0 = time.perf_counter()
        s.set_state(lad.w0, lad.T)
        s.sync()
        t1 = time.perf_counter()
        chunks = []
        while s.steps_done() < lad.S - 1:
            ta = time.perf_counter()
            s.run(chunk)
            s.sync()
            chunks.append(round((time.perf_counter() - ta) * 1e3, 3))
        out["runs"].append({"set_state_ms": round((t1 - t0) * 1e3, 3), "chunks_ms": chunks, "total_ms": round((time.perf_counter() - t0) * 1e3, 3)})
    # and back to back without intermediate syncs
    for run in range(3):
        t0 = time.perf_counter()
        s.set_state(lad.w0, lad.T)
        t1 = time.perf_counter()
        s.run(-1)
        t2 = time.perf_counter()
        s.sync()
        t3 = time.perf_counter()
        out["runs"].append({"set_state_ms": round((t1 - t0) * 1e3, 3), "enqueue_ms": round((t2 - t1) * 1e3, 3), "wait_ms": round((t3 - t2) * 1e3, 3)})
    n, ms = s.kernel_time()
    out["kernel_avg_ms"] = ms / max(n, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        wl = sys.argv[1] if len(sys.argv) > 1 else "ionosphere256"
        chunk = sys.argv[2] if len(sys.argv) > 2 else "1000"
        for stride in ("1", "0"):
            env = dict(os.environ, PTNN_TIMING_STRIDE=stride)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", wl, chunk], env=env, check=True)
