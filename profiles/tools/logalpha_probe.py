"""GPU probe: fp32 error of log alpha along the fixture trajectories (kernel's own log alpha vs the float64 oracle)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity
from parity import orc
d = parity.datasets()
out = {}
TRAJ = ["reg_rw", "reg_lg", "reg_lg_mackey", "cls_rw", "cls_lg", "cls_rw_ions", "reg_rw_noswitch", "reg_lg_wide", "cls_lg_wide"]
for key in TRAJ:
    g = parity.golden(f"trajectory_{key}.npz")
    topo = tuple(int(v) for v in g["topology"])
    task, S, gid, seed = int(g["task"]), int(g["S"]), int(g["gid"]), int(g["seed"])
    dname = str(g["dataset"])
    train, test = d[dname + "_train"], d[dname + "_test"]
    w0 = g["w0"].astype(np.float32)
    tape = orc.PhiloxTape(seed)
    rep = orc.Replica(task, topo, train, test, w0.astype(np.float64), float(g["T"]), S, bool(g["use_lg"]), 0.5, float(g["lr"]), tape, gid)
    la, lu, sc = np.zeros(S), np.zeros(S), np.zeros(S)
    for i in range(S - 1):
        rep.step(i)
        la[i], lu[i], sc[i] = rep.last_logalpha, np.log(rep.last_u), rep.last_scale
    s = parity.make_sampler(task, topo, train, test, R_local=1, R_global=8, first=gid, S=S, si=10 * S, use_lg=bool(g["use_lg"]),
                            lr=float(g["lr"]), seed=seed)
    s.set_state(w0[None, :], np.array([float(g["T"])], dtype=np.float32))
    while s.steps_done() < S - 1:
        s.run_segment()
    s.sync()
    tr = s.traces()
    lag = s.log_alpha()[0]
    acc_g = tr["accept"][0].astype(np.int64); acc_o = rep.accept_list.astype(np.int64)
    diff = np.nonzero(acc_g != acc_o)[0]
    first = int(diff[0]) - 2 if diff.size else None      # step whose decision differed
    upto = (S - 1) if first is None else first + 1
    err = np.abs(lag[:upto] - la[:upto])
    fin = np.isfinite(err)
    out[key] = dict(S=S, topo=topo, first_divergence_step=first, max_err=float(err[fin].max()), max_err_over_scale=float((err[fin] / sc[:upto][fin]).max()),
                    median_err=float(np.median(err[fin])), scale_median=float(np.median(sc[:upto])),
                    at_div=None if first is None else dict(la_o=la[first], la_g=float(lag[first]), lu=lu[first], scale=sc[first]),
                    describe=s.describe()["kernel"])
    print(key, out[key], flush=True)
    s.close()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "logalpha_probe.json"), "w"), indent=1, default=float)
