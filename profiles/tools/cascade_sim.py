"""Barrier vs pipelined swap cascade: rounds on the critical path of one whole run, from its accept trace (GPU).
   cascade_sim.py [workload]   -- what a persistent kernel that lets replica k start interval n+1 as soon as pairs (0,1)..(k,k+1)
   of round n are decided could gain over the synchronous swap barrier."""
import os, sys, argparse, numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_)
import bench
wlname = sys.argv[1] if len(sys.argv) > 1 else "sunspot64"
a = argparse.Namespace(waves=0, schedule=0, groups=0, bf16=False)
wl = dict(bench.WORKLOADS[wlname])
train, test, _ = bench.load_data(wl["data"])
lad = bench.Ladder(wl, a, train, test, 0, 1, 0)
s = lad.s
lad.whole_run()
acc = s.traces(pos_w=False)["accept"].astype(np.int64)
si, slots = wl["si"], s.describe()["slots_per_round"]
flags = np.diff(acc, axis=1)[:, 1:]
n_int = flags.shape[1] // si
R = flags.shape[0]
rounds = np.zeros((n_int, R), dtype=np.int64)
for it in range(n_int):
    f = flags[:, 1 + it * si: 1 + (it + 1) * si]
    for k, row in enumerate(f):
        pos = n = 0
        while pos < row.shape[0]:
            hit = np.flatnonzero(row[pos:pos + slots])
            pos += (hit[0] + 1) if hit.size else slots
            n += 1
        rounds[it, k] = n
barrier = rounds.max(axis=1).sum()
# pipelined cascade: replica k starts interval n+1 once pairs (0,1) .. (k,k+1) of round n are decided = replicas 0..k+1 done
F = np.zeros(R)
for it in range(n_int):
    pm = np.maximum.accumulate(F)                       # prefix max of finish times
    start = np.empty(R)
    start[:-1] = pm[1:]                                  # max over j <= k+1
    start[-1] = pm[-1]
    F = start + rounds[it]
print(f"{wlname}: slots {slots}, intervals {n_int}, mean rounds {rounds.mean():.2f}, mean of max {rounds.max(axis=1).mean():.2f}")
print(f"critical path in rounds: barrier {barrier}, pipelined cascade {F.max():.0f} ({barrier / F.max():.3f}x), lower bound (busiest replica) {rounds.sum(axis=0).max()}")
# even/odd or any other re-association is not allowed: the cascade is sequential in k (REG:694-759)
