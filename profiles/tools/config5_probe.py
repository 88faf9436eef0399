import sys, os, time, numpy as np
R_=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import parity
from parity import orc
train, test = parity.synthetic_regression(1280, 1024, 32, 512, seed=5)
topo=(32,512,1); P=orc.num_param(topo)
for R, S, lg, bf in ((128, 60, True, 0), (128, 120, False, 0), (128, 120, False, 1), (128, 60, True, 1)):
    tape=orc.PhiloxTape(1)
    w0=(0.3*np.stack([tape.w_init(r,P) for r in range(R)])).astype(np.float32)
    s=parity.make_sampler(0, topo, train, test, R_local=R, R_global=R, first=0, S=S, si=20, use_lg=lg, lr=0.1, seed=1, forward_bf16=bf)
    s.set_state(w0, np.array(orc.temperature_ladder(R,2),dtype=np.float32))
    s.run(21); s.sync()
    t0=time.perf_counter(); s.run(-1); s.sync(); dt=time.perf_counter()-t0
    n=S-1-21
    st=s.state()
    print(f"config5 shape R={R} lg={lg} bf16={bf}: {R*n/dt:.0f} samples/s ({dt/n*1e3:.2f} ms per step of all replicas), acc {100*st['num_accepted'].mean()/S:.1f}%", flush=True)
    s.close()
