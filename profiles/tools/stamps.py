"""Per-phase cycles of a round of the multi-CU speculative kernel (`segment_spec_kernel`), diagnostic build
(profiles/tools/build_stamps.sh), whole runs: replica 0 / slot 0.   stamps.py [workload]   (default mackey64; sunspot64 takes
the same kernel with --schedule 2)"""
import os, sys, time, argparse, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTNN_LIBRARY"] = os.environ.get("STAMPS_LIB", os.path.join(R, "profiles/tools/libptnn_stamps.so"))
sys.path.insert(0, R)
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "mackey64"
wl = dict(bench.WORKLOADS[name])
train, test, _ = bench.load_data(wl["data"])
names = ["prologue", "tape", "gd recompute", "proposal+sweep", "eval+MH", "publish", "wait wg", "gather", "commit"]
for use_lg in (True, False):
    w = dict(wl, lg=use_lg)
    lad = bench.Ladder(w, argparse.Namespace(waves=4, schedule=2, groups=4, bf16=False, shared_noise=1), train, test, 0, 1, 0)
    s = lad.s
    lad.whole_run(); s.debug_stamps()
    t0 = time.perf_counter(); lad.whole_run(); dt = time.perf_counter() - t0
    st = s.debug_stamps()
    nint = wl["S"] // wl["si"]
    rounds = st[9]; tot = sum(st[:9])
    print(f"{name} lg={use_lg} {s.describe()['kernel']}: {dt*1e3/nint:.3f} ms/interval, replica 0 slot 0: {rounds/nint:.1f} rounds/interval, {tot/nint/2.4e3:.1f} us/interval stamped (at 2.4 GHz)")
    for n, v in zip(names, st[:9]):
        print(f"    {n:16s} {v/max(rounds,1):9.0f} cyc/round  {100*v/tot:5.1f} %")
    s.close()
