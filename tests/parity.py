"""Shared helpers of the GPU parity tests and of __graft_entry__.smoke(): run the HIP path through the C ABI and the
CPU oracle on the same seeded inputs and compare.  The oracle is the checker here, never the product."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import ptnn_oracle as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

# fp32 kernel vs float64 oracle: unit values rel 1e-5 / abs 1e-6, log-likelihood sums rel 1e-5 (+ abs 2e-3
# on sums of several hundred terms), MH decisions identical until |log alpha - log u| is inside the fp32 noise
RTOL, ATOL = 2e-5, 2e-6
LOGALPHA_SLACK = 5e-3


def datasets():
    return dict(np.load(os.path.join(GOLDEN, "datasets.npz")))


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def make_sampler(task, topo, train, test, *, R_local, R_global, first, S, si, use_lg, lr, seed, l_prob=0.5, waves=0,
                 schedule=0, groups=0, trace_capacity=0, forward_bf16=0, swap_rule=0, shared_noise=0):
    import ptnn_amd
    from ptnn_amd import _lib
    pt = S * 0.6
    s = _lib.Sampler(device_id=0, task=task, n_in=topo[0], n_hidden=topo[1], n_out=topo[2], n_replicas_local=R_local,
                     n_replicas_global=R_global, first_global_replica=first, n_samples=S, swap_interval=si,
                     pt_switch_step=int(pt) if pt == int(pt) else -1, use_langevin=int(bool(use_lg)),
                     waves_per_replica=waves, schedule=schedule, groups_per_replica=groups, trace_capacity=trace_capacity, forward_bf16=forward_bf16, swap_rule=swap_rule, shared_noise=int(shared_noise), l_prob=l_prob, learn_rate=lr, step_w=0.025, step_eta=0.2,
                     sigma_squared=25.0, nu_1=0.0, nu_2=0.0, seed=seed)
    s.set_data(train, test)
    return s


def compare_replica_trace(tr, r, rep, label="", limit=None):
    """tr: Sampler.traces() dict; rep: oracle Replica run over the same tape.  Returns the first step whose MH
    decision differs (or None).  Everything before it must agree within the fp32 tolerances.  `limit`: compare only the first
    `limit` rows (the chains are known to part there for another reason, e.g. a near-tie decision in another replica whose
    state arrived through an exchange)."""
    S = rep.S if limit is None else min(rep.S, int(limit))
    acc_g = tr["accept"][r].astype(np.int64)[:S]
    acc_o = rep.accept_list.astype(np.int64)[:S]
    diff = np.nonzero(acc_g != acc_o)[0]
    first = int(diff[0]) if diff.size else None          # index into accept_list: decision of step first-2 differed
    upto = S if first is None else first - 1             # rows [0, upto) were produced by identical decisions
    if upto > 1:
        np.testing.assert_allclose(tr["pos_w"][r, :upto], rep.pos_w[:upto], rtol=RTOL, atol=2e-5, err_msg=label + " pos_w")
        lk = rep.likeh[:upto, 0]
        np.testing.assert_allclose(tr["likeh"][r, :upto], lk, rtol=5e-5, atol=5e-3, err_msg=label + " likeh")
        for nm in ("rmse_train", "rmse_test"):
            np.testing.assert_allclose(tr[nm][r, :upto], getattr(rep, nm)[:upto], rtol=1e-4, atol=1e-6, err_msg=label + nm)
        for nm in ("acc_train", "acc_test"):
            np.testing.assert_allclose(tr[nm][r, :upto], getattr(rep, nm)[:upto], rtol=1e-5, atol=1e-4, err_msg=label + nm)
    return first


class OracleRun:
    """Oracle replicas advanced step by step, recording log alpha and u of every step (for divergence analysis)."""

    def __init__(self, pt):
        self.pt = pt
        self.logalpha = np.full((pt.R, pt.S), np.nan)
        self.logu = np.full((pt.R, pt.S), np.nan)

    def run(self):
        pt = self.pt
        for i in range(pt.S - 1):
            for r, rep in enumerate(pt.replicas):
                rep.step(i)
                self.logalpha[r, i] = rep.last_logalpha
                self.logu[r, i] = np.log(rep.last_u)
            if orc.swap_trigger(pt.task, i, pt.si):
                pt.swap_round()
        rounds = int(pt.S / pt.si) if pt.si > 0 else 0
        if pt.swap_rule == 0 and rounds > pt.rounds_done:
            pt.swap_round(L=[rep.likelihood for rep in pt.replicas], apply=False)
        return self


def run_smoke_check():
    """Sunspot, FNN 4-5-1, 4 replicas, Langevin p=0.5, S=60, swaps every 10: HIP vs oracle on the same tape."""
    ds = datasets()
    train, test = ds["sunspot_train"], ds["sunspot_test"]
    topo, R, S, si, seed = (4, 5, 1), 4, 60, 10, 77
    pt = orc.PTOracle(orc.TASK_REG, topo, train, test, R, 2, R * S, si, use_lg=True, l_prob=0.5, lr=0.1, seed=seed)
    w0 = np.stack([rep.w for rep in pt.replicas]).astype(np.float32)
    for rep, w in zip(pt.replicas, w0):                     # both sides start from the same fp32-representable weights
        rep.__init__(orc.TASK_REG, topo, pt.train, pt.test, w.astype(np.float64), rep.T, S, True, 0.5, 0.1, pt.tape, rep.gid)
    o = OracleRun(pt).run()
    s = make_sampler(orc.TASK_REG, topo, train, test, R_local=R, R_global=R, first=0, S=S, si=si, use_lg=True, lr=0.1,
                     seed=seed)
    s.set_state(w0, np.array(pt.temperatures, dtype=np.float32))
    s.run(-1)
    s.sync()
    tr = s.traces()
    firsts = [compare_replica_trace(tr, r, pt.replicas[r], f"smoke r{r} ") for r in range(R)]
    nsw, tot, rounds = s.swap_stats()
    assert tot == pt.total_swap_proposals and rounds == pt.rounds_done, (tot, rounds)
    for r, f in enumerate(firsts):
        if f is not None:
            i = f - 2
            gap = abs(o.logalpha[r, i] - o.logu[r, i])
            assert gap < LOGALPHA_SLACK or np.isnan(gap), f"replica {r} diverged at step {i} with |log a - log u| = {gap}"
    s.close()
    return firsts


def synthetic_regression(n_rows, n_train, n_in, n_hidden, seed):
    """SURVEY.md 8(d) config-5 recipe (same as tests/golden/make_fixtures.py): X ~ U(0,1), teacher FNN run through the
    reference forward rule, y = clip(teacher + N(0, 0.02^2), 0, 1)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (n_rows, n_in))
    topo = (n_in, n_hidden, 1)
    w_t = np.concatenate([rng.standard_normal(n_in * n_hidden) / np.sqrt(n_in), rng.standard_normal(n_hidden) / np.sqrt(n_hidden),
                          rng.standard_normal(n_hidden) / np.sqrt(n_in), rng.standard_normal(1) / np.sqrt(n_hidden)])
    y = np.clip(orc.forward(X, w_t, topo)[1][:, 0] + rng.normal(0, 0.02, n_rows), 0, 1)
    data = np.hstack([X, y[:, None]])
    return data[:n_train], data[n_train:]
