"""Pins oracle/ptnn_oracle_c.c (the C restatement that follows the device through whole runs) against the SAME reference-generated
vectors as the numpy oracle (tests/golden/, made by make_fixtures.py importing REG/CLS), and against the numpy oracle itself.
CPU only; compiles the C file with gcc (seconds)."""
import glob
import os

import numpy as np
import pytest

import ptnn_oracle as orc
import ptnn_oracle_c as orc_c

from test_oracle_golden import DATA_OF, FUNC_CASES, SWAPTRAJ, _load

RTOL = 1e-10        # the C sums run in index order, numpy's pairwise: round-off apart


def test_tape_matches_numpy_tape():
    tape = orc.PhiloxTape(12345678901234567)
    for rep, step in ((0, 0), (3, 17), (63, 9998)):
        np.testing.assert_allclose(orc_c.step_scalars(tape.seed, rep, step), tape.step_scalars(rep, step), rtol=1e-14)
        np.testing.assert_allclose(orc_c.w_noise(tape.seed, rep, step, 31), tape.w_noise(rep, step, 31), rtol=1e-13, atol=1e-15)
        assert orc_c.step_scalars(tape.seed, rep, step)[:2] == tape.step_scalars(rep, step)[:2]      # the uniforms bit for bit


@pytest.mark.parametrize("key", FUNC_CASES)
def test_functions_against_reference_vectors(golden_dir, datasets, key):
    g = _load(golden_dir, f"functions_{key}.npz")
    topo = tuple(int(v) for v in g["topology"])
    task = int(g["task"])
    train, test = datasets[DATA_OF[key] + "_train"], datasets[DATA_OF[key] + "_test"]
    for wi in range(3):
        w = g[f"w{wi}"]
        for lr in (0.1, 0.01):
            np.testing.assert_allclose(orc_c.langevin_gradient(train, w, topo, lr, task), g[f"lg{wi}_lr{lr}"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(orc_c.langevin_gradient(train[:1], w, topo, lr, task), g[f"lg1row{wi}_lr{lr}"], rtol=RTOL, atol=1e-14)
            np.testing.assert_allclose(orc_c.langevin_gradient(train[::-1], w, topo, lr, task), g[f"lgperm{wi}_lr{lr}"], rtol=1e-9, atol=1e-12)
        for T in (1.0, 1.2599210498948732, 2.0, 10.0):
            if task == orc.TASK_REG:
                for tau in (0.01, 0.1):
                    l, fx, r, _ = orc_c.likelihood(task, topo, train, w, tau, T)
                    np.testing.assert_allclose([l, r], g[f"lik{wi}_T{T}_tau{tau}"], rtol=RTOL)
                    np.testing.assert_allclose(fx, g[f"fx{wi}"], rtol=RTOL)
                    l, fx, r, _ = orc_c.likelihood(task, topo, test, w, tau, T)
                    np.testing.assert_allclose([l, r], g[f"liktest{wi}_T{T}_tau{tau}"], rtol=RTOL)
                    np.testing.assert_allclose(orc_c.prior(task, topo, w, tau), g[f"prior{wi}_tau{tau}"], rtol=RTOL)
            else:
                l, fx, r, a = orc_c.likelihood(task, topo, train, w, 1.0, T)
                np.testing.assert_allclose([l, r, a], g[f"lik{wi}_T{T}"], rtol=RTOL)
                np.testing.assert_array_equal(fx, g[f"fx{wi}"])
                l, fx, r, a = orc_c.likelihood(task, topo, test, w, 1.0, T)
                np.testing.assert_allclose([l, r, a], g[f"liktest{wi}_T{T}"], rtol=RTOL)
                np.testing.assert_allclose(orc_c.prior(task, topo, w), g[f"prior{wi}"], rtol=RTOL)


TRAJ = sorted(os.path.basename(p)[len("trajectory_"):-4]
              for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "trajectory_*.npz")))


@pytest.mark.parametrize("key", TRAJ)
def test_single_replica_trajectory(golden_dir, datasets, key):
    """F4: ptReplica.run() of the reference itself under the Philox tape; every MH decision identical."""
    g = _load(golden_dir, f"trajectory_{key}.npz")
    topo = tuple(int(v) for v in g["topology"])
    task, S = int(g["task"]), int(g["S"])
    dname = str(g["dataset"])
    rep = orc_c.CReplica(task, topo, datasets[dname + "_train"], datasets[dname + "_test"], g["w0"], float(g["T"]), S, bool(g["use_lg"]),
                         0.5, float(g["lr"]), int(g["seed"]), int(g["gid"]))
    rec = rep.run(0, S - 1)
    assert rec["natural"].sum() == rep.num_accepted
    np.testing.assert_array_equal(rep.accept_list, g["accept_list"])
    np.testing.assert_allclose(rep.pos_w, g["pos_w"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(rep.likeh, g["likeh"], rtol=1e-9, atol=1e-9)
    for nm in ("rmse_train", "rmse_test", "acc_train", "acc_test"):
        np.testing.assert_allclose(getattr(rep, nm), g[nm], rtol=1e-9, atol=1e-12)
    fin, P = g["final_param"], rep.P
    np.testing.assert_allclose(rep.w, fin[:P], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(rep.eta, fin[P], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(rep.likelihood, fin[P + 1], rtol=1e-9)
    assert rep.adapttemp == (rep.T if key.endswith("noswitch") else 1)


@pytest.mark.parametrize("key", SWAPTRAJ)
def test_full_pt_run_with_swaps(golden_dir, datasets, key):
    """F6: multi-process run_chains() of the reference (stale likelihood, trigger index, phantom round)."""
    g = _load(golden_dir, f"swap_trajectory_{key}.npz")
    topo = tuple(int(v) for v in g["topology"])
    task, dname, R = int(g["task"]), str(g["dataset"]), int(g["R"])
    pt = orc.PTOracle(task, topo, datasets[dname + "_train"], datasets[dname + "_test"], R, int(g["maxtemp"]), int(g["NumSample"]),
                      int(g["si"]), use_lg=bool(g["use_lg"]), l_prob=0.5, lr=float(g["lr"]), seed=int(g["seed"]))
    orc_c.adopt(pt)
    pt.run()
    assert pt.num_swap == int(g["num_swap"]) and pt.total_swap_proposals == int(g["total_swap_proposals"])
    for r, rep in enumerate(pt.replicas):
        np.testing.assert_array_equal(rep.accept_list, g[f"accept_list_{r}"])
        np.testing.assert_allclose(rep.pos_w, g[f"pos_w_{r}"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(rep.likeh, g[f"likeh_{r}"], rtol=1e-9, atol=1e-9)
        for nm in ("rmse_train", "rmse_test", "acc_train", "acc_test"):
            np.testing.assert_allclose(getattr(rep, nm), g[f"{nm}_{r}"], rtol=1e-9, atol=1e-12)


def test_forced_decisions_are_reported_not_hidden():
    """The follow mode: imposing the chain's own decisions changes nothing; imposing the opposite is flagged on that step."""
    d = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "datasets.npz")))
    args = (orc.TASK_REG, (4, 5, 1), d["sunspot_train"], d["sunspot_test"])
    w0 = orc.PhiloxTape(9).w_init(0, 31)
    a = orc_c.CReplica(*args, w0, 1.3, 80, True, 0.5, 0.1, 9, 0)
    ra = a.run(0, 79)
    b = orc_c.CReplica(*args, w0, 1.3, 80, True, 0.5, 0.1, 9, 0)
    rb = b.run(0, 79, force=ra["natural"])
    np.testing.assert_array_equal(a.pos_w, b.pos_w)
    assert (rb["natural"] == ra["natural"]).all()
    c = orc_c.CReplica(*args, w0, 1.3, 80, True, 0.5, 0.1, 9, 0)
    flipped = ra["natural"].copy()
    flipped[5] ^= 1
    rc = c.run(0, 6, force=flipped[:6])
    assert rc["natural"][5] == ra["natural"][5] and c.last_forced and c.num_accepted == ra["natural"][:6].sum() + (1 if flipped[5] else -1)
    # the numpy oracle reports the same
    n = orc.Replica(*args, w0, 1.3, 80, True, 0.5, 0.1, orc.PhiloxTape(9), 0)
    for i in range(6):
        n.step(i, force=bool(flipped[i]))
    assert n.last_forced and n.num_accepted == c.num_accepted
    np.testing.assert_allclose(n.pos_w[:7], c.pos_w[:7], rtol=1e-12)


def test_set_state_continues_from_the_given_state():
    """CReplica.set_state / the sync rows of run(): the chain continues from the imposed (w, eta) with likelihood and prior
    re-evaluated there -- a chain synced to ITSELF (its own rows, rounded to float32 like a device's) stays within float32
    rounding of the free chain, and set_state reproduces what a fresh chain computes for that state."""
    d = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "datasets.npz")))
    args = (orc.TASK_REG, (4, 5, 1), d["sunspot_train"], d["sunspot_test"])
    w0 = orc.PhiloxTape(9).w_init(0, 31)
    a = orc_c.CReplica(*args, w0, 1.3, 80, True, 0.5, 0.1, 9, 0)
    ra = a.run(0, 79)
    b = orc_c.CReplica(*args, w0, 1.3, 80, True, 0.5, 0.1, 9, 0)
    rows = a.pos_w.astype(np.float32)
    rb = b.run(0, 79, force=ra["natural"], sync_w=rows[1:80])
    assert (rb["natural"] == ra["natural"]).all()
    np.testing.assert_allclose(b.pos_w, a.pos_w, rtol=0, atol=3e-7)          # one float32 rounding of a weight of size ~ 1
    np.testing.assert_allclose(rb["logalpha"], ra["logalpha"], rtol=0, atol=2e-6 * np.max(ra["scale"]))
    assert np.array_equal(b.w.astype(np.float32), rows[79])                    # the state IS the imposed row (last accepted one)
    # set_state against a fresh chain started in that state: the same likelihood / prior (fresh chains compute eta from the
    # residual variance, so hand it that eta)
    c = orc_c.CReplica(*args, b.w, 1.3, 80, True, 0.5, 0.1, 9, 0)
    c.adapttemp = b.adapttemp                                                  # past the switch at 0.6 S (R10)
    c.set_state(b.w, eta=b.eta)
    assert c.likelihood == pytest.approx(b.likelihood, rel=1e-13) and c.prior_current == pytest.approx(b.prior_current, rel=1e-13)


class _OracleAsDevice:
    """A finished oracle run dressed as a `_lib.Sampler` (traces, log alpha, swap log, counters): lets the follow-mode checker of
    tests/parity.py be exercised without a GPU."""

    def __init__(self, pt, logalpha):
        self.pt, self.la = pt, logalpha

    def traces(self):
        reps = self.pt.replicas
        return dict(pos_w=np.stack([r.pos_w for r in reps]).astype(np.float32), likeh=np.stack([r.likeh[:, 0] for r in reps]).astype(np.float32),
                    accept=np.stack([r.accept_list for r in reps]).astype(np.int32),
                    **{k: np.stack([getattr(r, k) for r in reps]).astype(np.float32) for k in ("rmse_train", "rmse_test", "acc_train", "acc_test")})

    def state(self):
        return dict(num_accepted=np.array([r.num_accepted for r in self.pt.replicas]))

    def log_alpha(self):
        return self.la.astype(np.float32)

    def swap_log(self):
        return np.array(self.pt.src_log, dtype=np.int32)

    def swap_stats(self):
        return self.pt.num_swap, self.pt.total_swap_proposals, self.pt.rounds_done


@pytest.mark.parametrize("task", [0, 1])
def test_follow_mode_checker_on_an_oracle_run(datasets, task):
    """follow_device_run (the whole-run comparison of the GPU tests) against a 'device' that IS an oracle run: nothing forced,
    every check passes; and after one MH decision of that 'device' is flipped far from a coin flip, the checker refuses it."""
    import parity
    topo, dname, lg, lr, mt = ((4, 5, 1), "sunspot", True, 0.1, 2) if task == 0 else ((4, 12, 3), "iris", False, 0.01, 10)
    R, S, si, seed = 5, 120, 10, 31
    args = (task, topo, datasets[dname + "_train"], datasets[dname + "_test"], R, mt, R * S, si)
    dev = orc_c.adopt(orc.PTOracle(*args, use_lg=lg, lr=lr, seed=seed))
    o = parity.OracleRun(dev).run()
    fake = _OracleAsDevice(dev, o.logalpha[:, :S - 1])
    rep = parity.follow_device_run(fake, fake.traces(), orc_c.adopt(orc.PTOracle(*args, use_lg=lg, lr=lr, seed=seed)), "self ")
    assert rep["forced_mh"] == 0 and rep["forced_swap_pairs"] == 0 and rep["steps"] == R * (S - 1) and rep["steps_over_bound"] == 0
    assert rep["synced"] and dev.rounds_done == int(S / si)
    # the drift form (decisions imposed, states free) on the same 'device'
    rep = parity.follow_device_run(fake, fake.traces(), orc_c.adopt(orc.PTOracle(*args, use_lg=lg, lr=lr, seed=seed)), "self ", sync=False)
    assert rep["forced_mh"] == 0 and rep["steps_over_bound"] == 0 and not rep["synced"]
    # a decision that is no coin flip: flip the accept counter trail of one chain at its most decided step
    tr = fake.traces()
    margin = np.abs(o.logalpha[2, :S - 2] - o.logu[2, :S - 2])
    i = int(np.nanargmax(np.where(np.isfinite(margin), margin, -1)))
    delta = 1 if tr["accept"][2, i + 2] == tr["accept"][2, i + 1] else -1
    tr["accept"][2, i + 2:] += delta
    fake.state = lambda: dict(num_accepted=np.array([r.num_accepted + (delta if k == 2 else 0) for k, r in enumerate(dev.replicas)]))
    with pytest.raises(AssertionError):
        parity.follow_device_run(fake, tr, orc_c.adopt(orc.PTOracle(*args, use_lg=lg, lr=lr, seed=seed)), "flipped ")
