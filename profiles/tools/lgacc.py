import os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import bench
train, test, _ = bench.load_sunspot()
S = 20002
s = bench.make_sampler(train, test, 64, 64, 0, S, 0, True, 2, 0, 0)
prev = None
for chunk in range(4):
    s.run(5000); s.sync()
    st = s.state()
    cur = (st["num_accepted"].astype(int), st["langevin_count"].astype(int), st["langevin_accepted"].astype(int))
    if prev is not None:
        d = [c - p for c, p in zip(cur, prev)]
        print(f"steps {chunk*5000}-{(chunk+1)*5000}: accepted {d[0].sum()} (per replica max {d[0].max()}), LG proposed {d[1].sum()}, LG accepted {d[2].sum()} -> a_LG = {100*d[2].sum()/max(d[1].sum(),1):.3f}% ; per replica LG acc:", d[2].tolist())
    prev = cur
