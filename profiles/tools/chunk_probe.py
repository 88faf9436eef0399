"""Kernel time of one whole run taken in 1 launch vs K launches (ptnn_run per chunk), per workload.   chunk_probe.py workload K"""
import os, sys, time, argparse
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench
name, K = sys.argv[1], int(sys.argv[2])
a = argparse.Namespace(waves=0, schedule=0, groups=0, bf16=False, shared_noise=1)
wl = dict(bench.WORKLOADS[name])
train, test, _ = bench.load_data(wl["data"])
lad = bench.Ladder(wl, a, train, test, 0, 1, 0)
s = lad.s
S, si = wl["S"], wl["si"]
from ptnn_amd.parallel_tempering import overlap_cuts
for chunks in (1, K, 1, K):
    ends = overlap_cuts(S, si, chunks)
    s.set_state(lad.w0, lad.T); s.sync(); s.kernel_time(reset=True)
    t0 = time.perf_counter(); done = 0
    for b in ends:
        s.run(-1 if b == S - 1 else b - done); done = b
    s.sync(); dt = time.perf_counter() - t0
    n, ms = s.kernel_time(reset=True)
    print(f"{name} {s.describe()['kernel']} chunks {len(ends)}: wall {dt*1e3:.2f} ms, {n} timed launches, kernel {ms:.2f} ms", flush=True)
s.close()
# a second handle in the same process: is the first run's extra time the handle's (memory first touched) or the process's (code loaded)?
lad = bench.Ladder(wl, a, train, test, 0, 1, 0)
s = lad.s
for k in range(3):
    s.set_state(lad.w0, lad.T); s.sync(); s.kernel_time(reset=True)
    t0 = time.perf_counter(); s.run(-1); s.sync(); dt = time.perf_counter() - t0
    n, ms = s.kernel_time(reset=True)
    print(f"second handle, run {k}: wall {dt*1e3:.2f} ms, kernel {ms:.2f} ms", flush=True)
s.close()
# the same K launches with the trace rows of each copied to the pinned images behind it (ptnn_trace_image_fetch), nothing else running
lad = bench.Ladder(wl, a, train, test, 0, 1, 0)
s = lad.s
s.set_state(lad.w0, lad.T)
s.trace_image()
for k in range(3):
    ends = overlap_cuts(S, si, K)
    s.set_state(lad.w0, lad.T); s.sync(); s.kernel_time(reset=True)
    t0 = time.perf_counter(); done = 0; row = 0; tk = []
    for b in ends:
        s.run(-1 if b == S - 1 else b - done); done = b
        tk.append(s.trace_fetch(row, b + 1 - row)); row = b + 1
    landed = []
    for t in tk:
        s.trace_wait(t); landed.append(round((time.perf_counter() - t0) * 1e3, 2))
    s.sync(); dt = time.perf_counter() - t0
    n, ms = s.kernel_time(reset=True)
    print(f"with copies, run {k}: wall {dt*1e3:.2f} ms, kernel {ms:.2f} ms, rows landed at {landed}", flush=True)
s.close()
