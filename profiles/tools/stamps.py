import os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTNN_LIBRARY"] = os.path.join(R, "profiles/tools/libptnn_stamps.so")
sys.path.insert(0, R)
import bench
train, test, _ = bench.load_sunspot()
names = ["prologue","tape","gd recompute","proposal+sweep","eval+MH","publish","wait wg","gather","commit"]
for use_lg, waves, groups in ((True,4,4),(False,4,4)):
    S = 111*100+2
    s = bench.make_sampler(train, test, 64, 64, 0, S, 0, use_lg, 2, waves, groups)
    s.run(10*100+1); s.sync(); s.debug_stamps()
    import time; t0=time.perf_counter(); s.run(100*100); s.sync(); dt=time.perf_counter()-t0
    st = s.debug_stamps()
    rounds = st[9]; tot = sum(st[:9])
    print(f"lg={use_lg} waves={waves} groups={groups}: {dt*1e3/100:.3f} ms/interval, replica 0 slot 0: {rounds/100:.1f} rounds/interval, {tot/100/2.4e3:.1f} us/interval stamped")
    for n, v in zip(names, st[:9]):
        print(f"    {n:16s} {v/max(rounds,1):9.0f} cyc/round  {100*v/tot:5.1f} %")
    print('   in-kernel clock: %.3f GHz (s_memtime / s_memrealtime x 100 MHz)' % (st[10]/max(st[11],1)*0.1))
    print('   ONE launch: span entry->last loop end %.1f us, longest prologue %.1f us, longest loop %.1f us' % ((st[15]-st[12])/100, st[13]/100, st[14]/100))
    per = np.array(st[16:16+128], dtype=np.float64).reshape(64,2)
    print('   per-replica us/interval:', np.round(per[:,0]/100/2.4e3).astype(int).tolist())
    print('   per-replica rounds/interval:', np.round(per[:,1]/100,1).tolist())
    s.close()
