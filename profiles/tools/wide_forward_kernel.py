"""model_wide_kernel on 256 weight vectors (one work-group per CU): its duration under rocprofv3 --kernel-trace --stats is ONE forward
pass of the config-5 net per work-group (+ staging 70 KB into LDS).  FWD=0 split operands (default), FWD=2 exact fp32 instruction."""
import os, sys, numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import bench, parity
from parity import orc
wl = dict(bench.WORKLOADS["synthetic512"])
train, test, _ = bench.load_data(wl["data"])
P = orc.num_param(wl["topo"])
rng = np.random.default_rng(1)
s = parity.make_sampler(0, wl["topo"], train, test, R_local=4, R_global=4, first=0, S=20, si=10, use_lg=False, lr=0.1, seed=3,
                        forward_bf16=int(os.environ.get("FWD", "0")))
w = (0.3 * rng.standard_normal((256, P))).astype(np.float32)
tau = np.full(256, 0.05, np.float32)
for _ in range(10):
    s.evaluate(w, tau)
s.close()
