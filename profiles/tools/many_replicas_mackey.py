"""Mackey-Glass 4-10-1 with many replicas on one GPU: packed schedule with 16-lane groups vs the multi-CU speculative one."""
import sys, os, time, numpy as np
R_=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import parity
from parity import orc
d = parity.datasets()
for R in (64, 256, 1024):
    for sched in (0, 2, 3):
        S, si = 1002, 100
        tape = orc.PhiloxTape(1)
        w0 = np.stack([tape.w_init(r, 61) for r in range(R)]).astype(np.float32)
        s = parity.make_sampler(0, (4,10,1), d["mackey_train"], d["mackey_test"], R_local=R, R_global=R, first=0, S=S, si=si, use_lg=True, lr=0.1, seed=1, schedule=sched)
        s.set_state(w0, np.array(orc.temperature_ladder(R, 2), dtype=np.float32))
        s.run(101); s.sync()
        t0=time.perf_counter(); s.run(-1); s.sync(); dt=time.perf_counter()-t0
        print(f"R={R} schedule={sched}: {R*(S-1-101)/dt/1e6:.2f} M samples/s", flush=True)
        s.close()
