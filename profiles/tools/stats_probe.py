import sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import parity
from parity import orc
d = parity.datasets()
for key, task, topo, name, lg, lr, maxtemp in (("sunspot_rw_r8",0,(4,5,1),"sunspot",False,0.1,2),("sunspot_lg_r8",0,(4,5,1),"sunspot",True,0.1,2),("iris_rw_r8",1,(4,12,3),"iris",False,0.01,10)):
    f = json.load(open(os.path.join(parity.GOLDEN, f"stats_{key}.json")))
    R, S, si = f["R"], f["S"], f["swap_interval"]
    P = orc.num_param(topo)
    print(key, "ref accept mean", np.mean([r["accept_pct"] for r in f["runs"]]), "swap", [round(r["swap_perc"],1) for r in f["runs"]], "rmse", [round(r["rmse_train_mean"],3) for r in f["runs"]])
    for seed in range(1, 9):
        s = parity.make_sampler(task, topo, d[name+"_train"], d[name+"_test"], R_local=R, R_global=R, first=0, S=S, si=si, use_lg=lg, lr=lr, seed=seed)
        tape = orc.PhiloxTape(seed)
        w0 = np.stack([tape.w_init(r, P) for r in range(R)]).astype(np.float32)
        s.set_state(w0, np.array(orc.temperature_ladder(R, maxtemp), dtype=np.float32))
        s.run(-1); s.sync()
        tr = s.traces(pos_w=False)
        st = s.state(); nsw, tot, rounds = s.swap_stats()
        b = S//2
        print("  gpu seed", seed, "swap %.1f" % (100*nsw/tot), "acc", [round(100*a/S,1) for a in st["num_accepted"]], "rmse %.4f %.4f" % (tr["rmse_train"][:,b:].mean(), tr["rmse_test"][:,b:].mean()), "acc_tr %.1f" % tr["acc_train"][:,b:].mean())
        s.close()
