import sys, os, time, numpy as np
R_=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import parity
from parity import orc
train, test = parity.synthetic_regression(1280, 1024, 32, 512, seed=5)
topo=(32,512,1); P=orc.num_param(topo)
s=parity.make_sampler(0, topo, train, test, R_local=2, R_global=2, first=0, S=10, si=100, use_lg=True, lr=0.1, seed=1)
tape=orc.PhiloxTape(1)
for n in (1, 128):
    W=(0.3*np.stack([tape.w_init(r,P) for r in range(n)])).astype(np.float32)
    s.langevin_gradient(W); s.evaluate(W, 0.01)
    t0=time.perf_counter(); s.langevin_gradient(W); t1=time.perf_counter(); s.evaluate(W, 0.01); t2=time.perf_counter()
    print(f"n={n}: langevin_gradient call {1e6*(t1-t0):.0f} us, evaluate call {1e6*(t2-t1):.0f} us (incl. H2D/D2H of {n} x 70 KB)")
s.close()
