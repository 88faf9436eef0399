"""Per-phase cycles of a round of the packed speculative kernel (diagnostic build -DPTNN_STAMPS: profiles/tools/build_stamps.sh),
Sunspot 64 replicas, Langevin p = 0.5, one whole run of S = 10 000: replica 0 / wave 0."""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTNN_LIBRARY"] = os.path.join(R, "profiles/tools/libptnn_stamps.so")
sys.path.insert(0, R)
import bench
import argparse
a = argparse.Namespace(waves=0, schedule=int(os.environ.get("SCHED", "3")), groups=int(os.environ.get("GROUPS", "0")), bf16=False, shared_noise=1)
wl = dict(bench.WORKLOADS[os.environ.get("WORKLOAD", "sunspot64")])
train, test, _ = bench.load_data(wl["data"])
lad = bench.Ladder(wl, a, train, test, 0, 1, 0)
s = lad.s
lad.whole_run(); s.debug_stamps()
t0 = time.perf_counter(); lad.whole_run(); dt = time.perf_counter() - t0
st = s.debug_stamps()
names = ["loop head", "tape", "proposal", "sweep", "wait for forward passes", "MH", "commit"]
rounds = st[9]; tot = sum(st[:9]); nint = wl["S"] // wl["si"]
print(f"{s.describe()['kernel']}: {dt*1e3/nint:.3f} ms/interval, replica 0 wave 0: {rounds/nint:.1f} rounds/interval, {tot/nint/2.4e3:.1f} us/interval stamped (at 2.4 GHz)")
for n, v in zip(names, st[:7]):
    print(f"    {n:26s} {v/max(rounds,1):9.0f} cyc/round  {100*v/tot:5.1f} %")
print(f"    forward passes on wave 2: {st[11]/max(rounds,1):9.0f} cyc/round")
per = np.array(st[16:16+128], dtype=np.float64).reshape(64, 2)
print('   per-replica us/interval:', np.round(per[:, 0]/nint/2.4e3).astype(int).tolist())
print('   per-replica rounds/interval:', np.round(per[:, 1]/nint, 1).tolist())
print('   per-replica us/round:', np.round(per[:, 0]/np.maximum(per[:, 1], 1)/2.4e3, 1).tolist())
s.close()
