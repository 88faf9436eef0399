"""Per-phase cycles of one round of the prefetching-tree kernel (diagnostic build: build_stamps.sh 1 4 3), Iris 16 replicas,
one whole run: root work-group of replica 0, wave 0.   stamps_tree.py [workload]"""
import os, sys, time, argparse
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTNN_LIBRARY"] = os.environ.get("STAMPS_LIB", os.path.join(R, "profiles/tools/libptnn_stamps.so"))
sys.path.insert(0, R)
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "iris16"
a = argparse.Namespace(waves=0, schedule=0, groups=0, bf16=False, shared_noise=1)
wl = dict(bench.WORKLOADS[name])
train, test, _ = bench.load_data(wl["data"])
lad = bench.Ladder(wl, a, train, test, 0, 1, 0)
s = lad.s
lad.whole_run(); s.debug_stamps()
t0 = time.perf_counter(); lad.whole_run(); dt = time.perf_counter() - t0
st = s.debug_stamps()
names = ["loop head / switch", "tapes (not drawn ahead)", "proposal + forward image", "forward pass + likelihood", "publish + next tapes", "wait for all records",
         "decisions", "state rebuild + trace rows"]
rounds = st[9]; tot = sum(st[:9])
d = s.describe()
print(f"{name} {d['kernel']} {d['schedule']} G={d['groups_per_replica']}: {dt*1e3:.2f} ms/run = {wl['R']*wl['S']/dt/1e6:.2f} M samples/s; root group of replica 0: "
      f"{rounds} rounds, {tot/max(rounds,1):.0f} ticks per round (s_memtime), in-kernel total {st[10]} ticks")
for n, v in zip(names, st[:8]):
    print(f"    {n:32s} {v/max(rounds,1):9.0f} ticks/round  {100*v/max(tot,1):5.1f} %")
s.close()
