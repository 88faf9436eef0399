import sys, os, time, json, numpy as np
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import parity
from parity import orc
d = parity.datasets()
CFG = [("iris16_rw",1,(4,12,3),"iris",False,0.01,16,10,2000,40),
       ("iris16_lg",1,(4,12,3),"iris",True,0.01,16,10,2000,40),
       ("mackey64_lg",0,(4,10,1),"mackey",True,0.1,64,2,2000,100),
       ("ions256_rw",1,(34,50,2),"ions",False,0.01,256,10,600,100),
       ("ions32_lg",1,(34,50,2),"ions",True,0.01,32,10,300,100),
       ("sunspot64_lg",0,(4,5,1),"sunspot",True,0.1,64,2,2000,100)]
for key, task, topo, name, lg, lr, Rr, mt, S, si in CFG:
    P = orc.num_param(topo)
    tape = orc.PhiloxTape(1)
    w0 = np.stack([tape.w_init(r, P) for r in range(Rr)]).astype(np.float32)
    T = np.array(orc.temperature_ladder(Rr, mt), dtype=np.float32)
    for sched, waves in ((1,0),(1,1),(2,0),(2,2),(2,4)):
        try:
            s = parity.make_sampler(task, topo, d[name+"_train"], d[name+"_test"], R_local=Rr, R_global=Rr, first=0, S=S, si=si, use_lg=lg, lr=lr, seed=1, schedule=sched, waves=waves)
        except Exception as e:
            print(key, sched, waves, "ERR", str(e)[:60]); continue
        s.set_state(w0, T)
        s.run(si+1); s.sync()
        t0=time.perf_counter(); s.run(-1); s.sync(); dt=time.perf_counter()-t0
        st=s.state(); nsw,tot,_=s.swap_stats()
        print(f"{key:14s} sched {sched} waves {waves}: {Rr*(S-2-si)/dt/1e6:8.3f} M samples/s  acc% {100*st['num_accepted'].mean()/S:5.1f} swap% {100*nsw/max(tot,1):5.1f}", flush=True)
        s.close()
