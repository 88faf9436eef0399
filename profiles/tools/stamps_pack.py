import os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTNN_LIBRARY"] = os.path.join(R, "profiles/tools/libptnn_stamps.so")
sys.path.insert(0, R)
import bench, time
train, test, _ = bench.load_sunspot()
names = ["loop head","tape","proposal","sweep","wait for forward passes","MH","commit"]
S = 111*100+2
s = bench.make_sampler(train, test, 64, 64, 0, S, 0, True, 3, 0, 0)
s.run(10*100+1); s.sync(); s.debug_stamps()
t0=time.perf_counter(); s.run(100*100); s.sync(); dt=time.perf_counter()-t0
st = s.debug_stamps()
rounds = st[9]; tot = sum(st[:9])
print(f"packed: {dt*1e3/100:.3f} ms/interval, replica 0 wave 0: {rounds/100:.1f} rounds/interval, {tot/100/2.4e3:.1f} us/interval stamped")
for n, v in zip(names, st[:7]):
    print(f"    {n:26s} {v/max(rounds,1):9.0f} cyc/round  {100*v/tot:5.1f} %")
print(f"    forward passes on wave 2: {st[11]/max(rounds,1):9.0f} cyc/round")
per = np.array(st[16:16+128], dtype=np.float64).reshape(64,2)
print('   per-replica us/interval:', np.round(per[:,0]/100/2.4e3).astype(int).tolist())
print('   per-replica rounds/interval:', np.round(per[:,1]/100,1).tolist())
print('   per-replica us/round:', np.round(per[:,0]/np.maximum(per[:,1],1)/2.4e3,1).tolist())
s.close()
