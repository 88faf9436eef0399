"""Per-phase cycles of one MH step of the cooperative kernel (diagnostic build: build_stamps.sh 1 34 2 / 1 4 3)."""
import os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PTNN_LIBRARY", os.path.join(R, "profiles/tools/libptnn_stamps.so"))
sys.path.insert(0, R)
import bench, time
from ptnn_amd import _lib, ladder, philox
wl = sys.argv[1] if len(sys.argv) > 1 else "ionosphere256"
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 0
task, topo, dname, Rn, use_lg, lr, maxtemp, si, desc = bench.OTHER_CONFIGS[wl]
d = np.load(os.path.join(R, "tests", "golden", "datasets.npz"))
train, test = d[dname + "_train"], d[dname + "_test"]
Pw = topo[0] * topo[1] + topo[1] * topo[2] + topo[1] + topo[2]
S = 40 * si + 2
s = _lib.Sampler(device_id=0, task=task, n_in=topo[0], n_hidden=topo[1], n_out=topo[2], n_replicas_local=Rn, n_replicas_global=Rn,
                 first_global_replica=0, n_samples=S, swap_interval=si, pt_switch_step=bench.switch_step(S), use_langevin=int(use_lg),
                 waves_per_replica=waves, schedule=1, groups_per_replica=0, l_prob=0.5, learn_rate=lr, step_w=0.025, step_eta=0.2,
                 sigma_squared=25.0, seed=bench.SEED, trace_capacity=0)
s.set_data(train, test)
s.set_state(np.stack([philox.initial_weights(bench.SEED, r, Pw) for r in range(Rn)]), ladder.temperatures(Rn, maxtemp))
s.run(5 * si); s.sync(); s.debug_stamps()
t0 = time.perf_counter(); s.run(20 * si); s.sync(); dt = time.perf_counter() - t0
st = s.debug_stamps()
names = ["loop head", "tape", "proposal", "forward image", "forward pass + likelihood", "prior, MH, update", "trace row"]
steps = st[9]; tot = sum(st[:9])
print(f"{wl} waves={waves}: {dt*1e3/20:.3f} ms/interval; replica 0 wave 0: {tot/max(steps,1):.0f} cycles per MH step")
for n, v in zip(names, st[:7]):
    print(f"    {n:28s} {v/max(steps,1):9.0f} cyc/step  {100*v/tot:5.1f} %")
s.close()
