"""How many speculative rounds does a swap interval take as a function of the window (slots per round)?  Runs one whole run of a
bench workload on the GPU, takes the MH decisions from the traces and replays the round structure on the host: a round commits the
prefix of its window up to and including the first accepted step; an interval lasts as long as its slowest replica.
    python profiles/tools/window_sim.py [workload]"""
import json, os, sys, argparse
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_)
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "mackey64"
wl = dict(bench.WORKLOADS[name])
a = argparse.Namespace(waves=0, schedule=0, groups=0, bf16=False, shared_noise=1)
train, test, _ = bench.load_data(wl["data"])
lad = bench.Ladder(wl, a, train, test, 0, 1, 0)
lad.whole_run(); lad.s.sync()
acc = lad.s.traces(pos_w=False)["accept"]                  # accept[r, i] = accepted steps before row i
flags = np.diff(acc.astype(np.int64), axis=1) > 0          # flags[r, i]: MH step i accepted  (row i + 1 follows step i)
R, n = flags.shape
si = wl["si"]
out = {"workload": name, "replicas": R, "steps": n, "accept_pct": float(100 * flags.mean())}
for W in (8, 16, 32, 64):
    tot_max, tot_mean = 0.0, 0.0
    nint = 0
    for b in range(0, n, si):
        e = min(b + si, n)
        rounds = np.zeros(R)
        for r in range(R):
            i, k = b, 0
            f = flags[r]
            while i < e:
                w = f[i:min(i + W, e)]
                hit = np.flatnonzero(w)
                i += (hit[0] + 1) if hit.size else w.size
                k += 1
            rounds[r] = k
        tot_max += rounds.max(); tot_mean += rounds.mean(); nint += 1
    out[f"window_{W}"] = {"rounds_per_interval_slowest": tot_max / nint, "rounds_per_interval_mean": tot_mean / nint}
print(json.dumps(out))
lad.s.close()
