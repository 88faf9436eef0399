"""One SGD epoch of the 32-512-1 net vs a PAIR of epochs through one row loop (ptnn_time_sgd_epoch, in-kernel counter)."""
import sys, os, numpy as np
R_=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import parity
from parity import orc
train, test = parity.synthetic_regression(1280, 1024, 32, 512, seed=5)
topo=(32,512,1); P=orc.num_param(topo)
s=parity.make_sampler(0, topo, train, test, R_local=2, R_global=2, first=0, S=10, si=100, use_lg=True, lr=0.1, seed=1)
w=(0.3*orc.PhiloxTape(1).w_init(0,P)).astype(np.float32)
one, pair = s.time_sgd_epoch(w, reps=20, pair=True)
print(f"one epoch {one*1e3:.1f} us ({one*1e6/1024*2.38:.0f} cycles per row at 2.38 GHz); a pair {pair*1e3:.1f} us = {pair/one:.2f} x one")
s.close()
