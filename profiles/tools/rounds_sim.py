import os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import bench
train, test, _ = bench.load_sunspot()
S = 11102
s = bench.make_sampler(train, test, 64, 64, 0, S, 0, True, 2, 0, 0)
s.run(-1); s.sync()
tr = s.traces(pos_w=False)
acc = tr["accept"].astype(np.int64)           # acc[r, i+1] = count before step i
flags = np.diff(acc, axis=1)[:, 1:]            # flags[r, i] = step i accepted (i = 0..S-3)
lgc = s.state()["langevin_count"]
def rounds_for(fl, k):
    pos, n, rounds = 0, len(fl), 0
    while pos < n:
        w = fl[pos:pos+k]
        hit = np.nonzero(w)[0]
        pos += (hit[0] + 1) if hit.size else len(w)
        rounds += 1
    return rounds
for k in (16, 32, 64):
    per = []
    for it in range(10, 110):                  # the bench's timed intervals: steps 1001..11000
        a, b = 1001 + (it - 10) * 100, 1001 + (it - 9) * 100
        per.append([rounds_for(flags[r, a:b], k) for r in range(64)])
    per = np.array(per)
    mx = per.max(axis=1)
    print(f"k={k}: rounds per interval: mean over replicas {per.mean():.2f}, replica 0 {per[:,0].mean():.2f}; max over replicas: mean {mx.mean():.2f} median {np.median(mx):.1f} min {mx.min()} max {mx.max()}; argmax histogram (top 8):", np.bincount(per.argmax(axis=1), minlength=64).argsort()[::-1][:8].tolist())
A = np.array([[flags[r, 1001 + i*100: 1101 + i*100].sum() for r in range(64)] for i in range(100)])
print("accepts per interval: mean over replicas %.2f, max over replicas mean %.2f, overall max %d" % (A.mean(), A.max(axis=1).mean(), A.max()))

# dataflow schedule: replica j may start interval t+1 once replicas 0..j+1 have finished interval t (the bubble pass decides
# pair (k,k+1) from positions <= k+1 only).  Longest path through that DAG vs the synchronous barrier, in rounds of 16 slots.
per = np.array([[rounds_for(flags[r, 1001 + it * 100: 1101 + it * 100], 16) for r in range(64)] for it in range(100)], dtype=np.float64)
F = np.zeros(64)
for t in range(100):
    M = np.maximum.accumulate(F)                       # M[j] = max_{i<=j} F[i]
    need = np.concatenate([M[1:], M[-1:]])             # max over 0..j+1
    F = need + per[t]
print("k=16: barrier schedule %.1f rounds per interval (sum of per-interval maxima / 100); dataflow schedule %.1f; replica 0 alone %.1f; mean replica %.1f"
      % (per.max(axis=1).sum() / 100, F.max() / 100, per[:, 0].sum() / 100, per.mean()))
for lag in (1, 2, 4):
    # variant: swap decisions made on L that is `lag` intervals old is NOT the reference's algorithm; not evaluated
    pass
