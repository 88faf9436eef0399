"""Per-phase cycles of one MH step of the cooperative kernel (diagnostic build: build_stamps.sh 1 4 3 for iris16,
1 34 2 for ionosphere256), one whole run: replica 0 / wave 0.   stamps_coop.py [workload] [waves]"""
import os, sys, time, argparse
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTNN_LIBRARY"] = os.environ.get("STAMPS_LIB", os.path.join(R, "profiles/tools/libptnn_stamps.so"))
sys.path.insert(0, R)
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "iris16"
a = argparse.Namespace(waves=int(sys.argv[2]) if len(sys.argv) > 2 else 0, schedule=1, groups=0, bf16=False, shared_noise=1)
wl = dict(bench.WORKLOADS[name])
train, test, _ = bench.load_data(wl["data"])
lad = bench.Ladder(wl, a, train, test, 0, 1, 0)
s = lad.s
lad.whole_run(); s.debug_stamps()
t0 = time.perf_counter(); lad.whole_run(); dt = time.perf_counter() - t0
st = s.debug_stamps()
names = ["loop head", "switch-step re-evaluation", "proposal + forward image (+ SGD epochs)", "|proposal|^2 part, next tape", "forward pass + likelihood", "prior, MH, buffer rotation", "trace row"]
steps = st[9]; tot = sum(st[:9])
d = s.describe()
print(f"{name} {d['kernel']} {d}: {dt*1e3:.2f} ms/run = {wl['R']*wl['S']/dt/1e6:.2f} M samples/s; replica 0 wave 0: {tot/max(steps,1):.0f} cycles per MH step")
for n, v in zip(names, st[:7]):
    print(f"    {n:44s} {v/max(steps,1):9.0f} cyc/step  {100*v/tot:5.1f} %")
fw = st[150:154]
if sum(fw):
    for n, v in zip(["matrix products + epilogues (wave 0)", "waiting for the other waves", "scoring the rows", "work-group reduction"], fw):
        print(f"      forward: {n:38s} {v/max(steps,1):9.0f} cyc/step")
s.close()
