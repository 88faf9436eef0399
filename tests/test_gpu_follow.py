"""Whole runs FOLLOWED by the oracle to their last step (tests/parity.py: follow_device_run).

The prefix comparisons of test_gpu_parity.py stop at the first MH or cascade decision the fp32 kernel and the float64 oracle take
differently (inside the fp32 error of log alpha): from there on the two chains are different chains.  Here the oracle -- the C
restatement oracle/ptnn_oracle_c.c, pinned to the same reference-generated vectors as the numpy one -- is advanced with the
DEVICE's decisions imposed, so its inputs stay comparable to the end of the run and every step of every chain is held to the
log-alpha bound, every trace row to a tolerance, every differing decision to the coin-flip bound: the ring wrap of the tape, the
cached gradient after cross-slot moves, the temperature switch at 0.6 S, the phantom round -- all of it against the oracle, at
the BASELINE replica counts and, for the metric's own configuration, at the bench's exact size (64 chains x 10 000 samples)."""
import json
import os

import numpy as np
import pytest

import parity
from parity import orc

pytestmark = pytest.mark.gpu

import ptnn_oracle_c as orc_c  # noqa: E402


def ds():
    return parity.datasets()


def followed_run(task, topo, dname, R, lg, lr, maxtemp, S, si, seed, label, shared_noise=0, w0=None, **sampler_kw):
    d = ds()
    train, test = d[dname + "_train"], d[dname + "_test"]
    if not os.environ.get("PTNN_FOLLOW_F64DATA"):
        # both sides read the SAME data: the values the device holds (ptnn_set_data takes float32), as it is done for the initial
        # weights -- the files' decimal values are not float32 numbers, and with tau^2 ~ 1e-4 their rounding alone moves a
        # log-likelihood by ~1e-2 (it cancels between a chain's proposal and its current state, not between two implementations)
        train, test = (np.asarray(a, dtype=np.float32).astype(np.float64) for a in (train, test))
    pt = orc.PTOracle(task, topo, train, test, R, maxtemp, R * S, si, use_lg=lg, l_prob=0.5, lr=lr, seed=seed, shared_noise=bool(shared_noise))
    if w0 is None:
        w0 = np.stack([rep.w for rep in pt.replicas])
    w0 = np.asarray(w0, dtype=np.float32)                   # both sides start from the same fp32-representable weights
    orc_c.adopt(pt, w0=w0.astype(np.float64))
    s = parity.make_sampler(task, topo, train, test, R_local=R, R_global=R, first=0, S=S, si=si, use_lg=lg, lr=lr, seed=seed,
                            shared_noise=shared_noise, **sampler_kw)
    s.set_state(w0, np.array(pt.temperatures, dtype=np.float32))
    s.run(-1)
    s.sync()
    tr = s.traces()
    report = parity.follow_device_run(s, tr, pt, label)      # decisions AND state imposed: every step held to the single-step bound
    report["schedule"] = s.describe()["schedule"]
    if os.environ.get("PTNN_FOLLOW_DRIFT"):
        # the drift form beside it (decisions imposed, the two states free to drift apart over the run): reported, held to its own
        # looser bound -- what the per-step form above removes from the comparison
        pt2 = orc.PTOracle(task, topo, train, test, R, maxtemp, R * S, si, use_lg=lg, l_prob=0.5, lr=lr, seed=seed, shared_noise=bool(shared_noise))
        orc_c.adopt(pt2, w0=w0.astype(np.float64))
        drift = parity.follow_device_run(s, tr, pt2, label + "drift ", sync=False)
        report["drift"] = {k: drift[k] for k in ("max_ratio_fresh", "max_ratio_stale", "max_abs_err", "forced_mh", "forced_swap_pairs",
                                                 "pos_w_abs_excess", "likeh_abs_excess")}
        report["state_drift_max"] = float(max(np.abs(np.asarray(tr["pos_w"][r], dtype=np.float64) - pt2.replicas[r].pos_w).max() for r in range(R)))
    s.close()
    print(label, json.dumps(report))
    if os.environ.get("PTNN_FOLLOW_REPORT"):
        with open(os.environ["PTNN_FOLLOW_REPORT"], "a") as f:
            f.write(json.dumps(dict(label=label.strip(), R=R, S=S, swap_interval=si, topology=list(topo), data=dname, **report)) + "\n")
    return report


# name: task, topology, data set, replicas, Langevin, lr, maxtemp, samples per replica, swap interval
WHOLE = {
    "sunspot64": (0, (4, 5, 1), "sunspot", 64, True, 0.1, 2, 1500, 100),
    "iris16": (1, (4, 12, 3), "iris", 16, False, 0.01, 10, 3000, 100),
    "mackey64": (0, (4, 10, 1), "mackey", 64, True, 0.1, 2, 1000, 100),
    "ionosphere256": (1, (34, 50, 2), "ions", 256, False, 0.01, 10, 400, 100),
    # two more problems of the reference's tables on their shipped data: Lazer (REG:881-917, 4-5-1, Langevin) and Cancer
    # (CLS:950-957: 9 inputs, 12 hidden units, 2 classes, random-walk), 10 chains as in its main()
    "lazer10": (0, (4, 5, 1), "lazer", 10, True, 0.1, 2, 2000, 20),
    "cancer10": (1, (9, 12, 2), "cancer", 10, False, 0.01, 10, 2000, 40),
    # BASELINE configs 1 and 3 with their literal five-input topologies (series re-embedded with window 6, tests/golden/make_fixtures.py)
    "sunspot5_r4": (0, (5, 5, 1), "sunspot5", 4, True, 0.1, 2, 5000, 50),
    "mackey5_r64": (0, (5, 10, 1), "mackey5", 64, True, 0.1, 2, 1000, 100),
}


@pytest.mark.parametrize("name", list(WHOLE))
def test_whole_run_followed_to_the_end(name):
    """BASELINE replica counts under the schedule the library picks, hundreds to thousands of steps per chain across the
    temperature switch and 4 - 30 swap rounds: every step, decision and trace row against the oracle."""
    task, topo, dname, R, lg, lr, maxtemp, S, si = WHOLE[name]
    rep = followed_run(task, topo, dname, R, lg, lr, maxtemp, S, si, 900 + R, f"{name} ")
    assert rep["steps"] == R * (S - 1)
    # decisions the two sides take differently are coin flips inside the fp32 bound: a handful per 10^5, not a trend
    assert rep["forced_mh"] <= max(3, rep["steps"] // 5000), rep
    assert rep["forced_swap_pairs"] <= max(2, rep["swap_pairs"] // 500), rep


def test_config5_at_its_bench_shape_followed_to_the_end():
    """BASELINE config 5 as bench.py runs it on one GPU -- FNN 32-512-1 (P = 17 409), 1024 / 256 rows, 128 chains, Langevin p = 0.5,
    the LDS-resident wide kernel with two work-groups per chain and compact traces -- for 13 samples per chain across two swap rounds
    (the float64 oracle needs 50 ms per step of this net even in C): every step, decision and trace row against the oracle."""
    train, test = parity.synthetic_regression(1280, 1024, 32, 512, seed=5)
    topo, R, S, si, seed = (32, 512, 1), 128, 13, 5, 1
    pt = orc.PTOracle(orc.TASK_REG, topo, train, test, R, 2, R * S, si, use_lg=True, l_prob=0.5, lr=0.1, seed=seed)
    w0 = (0.3 * np.stack([rep.w for rep in pt.replicas])).astype(np.float32)      # bench.py's scale for wide nets
    orc_c.adopt(pt, w0=w0.astype(np.float64))
    s = parity.make_sampler(orc.TASK_REG, topo, train, test, R_local=R, R_global=R, first=0, S=S, si=si, use_lg=True, lr=0.1, seed=seed)
    info = s.describe()
    assert info["lds_resident_state"] == 1 and info["compact_traces"] == 1 and info["groups_per_replica"] == 2, info
    s.set_state(w0, np.array(pt.temperatures, dtype=np.float32))
    s.run(-1)
    s.sync()
    tr = s.traces()
    rep = parity.follow_device_run(s, tr, pt, "config5 ", threads=16)
    s.close()
    print("config5", json.dumps(rep))
    assert rep["steps"] == R * (S - 1) and rep["forced_mh"] <= 2 and rep["forced_swap_pairs"] <= 2, rep


@pytest.mark.parametrize("shared_noise", [0, 1])
def test_headline_run_at_the_bench_size_followed_to_the_end(shared_noise):
    """The BASELINE metric's own run, exactly as bench.py makes it (Sunspot 4-5-1, 64 chains x 10 000 samples, Langevin p = 0.5,
    swap interval 100, seed 1, Philox stream-3 initial weights): 639 936 MH steps, 100 swap rounds, the switch at step 6000 and
    the phantom round -- followed by the oracle to the last step, under the default independent noise streams and under the
    reference's shared stream (Q14)."""
    import ptnn_amd  # noqa: F401
    from ptnn_amd import philox
    R, S, si, seed, topo = 64, 10000, 100, 1, (4, 5, 1)
    w0 = np.stack([philox.initial_weights(seed, r, 31) for r in range(R)])
    rep = followed_run(0, topo, "sunspot", R, True, 0.1, 2, S, si, seed, f"headline shared_noise={shared_noise} ", shared_noise=shared_noise, w0=w0)
    assert rep["steps"] == R * (S - 1) and rep["schedule"] == "packed-speculative"
    assert rep["forced_mh"] <= 64 and rep["forced_swap_pairs"] <= 16, rep


@pytest.mark.parametrize("case", ["reg_speculative", "reg_cooperative", "reg_packed_lg", "cls_cooperative"])
def test_nan_proposals_are_accepted_through_the_mh_branch(case):
    """R9 / Q8: `min(1, exp(nan))` is 1 in the reference (REG:372-376), so a proposal whose log alpha is NaN is ACCEPTED.  One
    NaN in the training data makes every likelihood NaN on both sides: every step of every chain must be accepted, the
    recorded weights must follow the proposals, and the chains must move through the swap rounds exactly as the oracle's."""
    d = ds()
    if case.startswith("reg"):
        task, topo, train, test, lr, mt = 0, (4, 5, 1), d["sunspot_train"].copy(), d["sunspot_test"], 0.1, 2
        train[7, 4] = np.nan                                 # one target: (y - fx)^2 is NaN, fx and the SGD epoch of other rows are not
    else:
        task, topo, train, test, lr, mt = 1, (4, 12, 3), d["iris_train"].copy(), d["iris_test"], 0.01, 10
        train[5, 2] = np.nan                                 # one input feature
    lg = case.endswith("_lg")
    sched = {"reg_speculative": 2, "reg_cooperative": 1, "reg_packed_lg": 3, "cls_cooperative": 1}[case]
    R, S, si, seed = 4, 40, 8, 404
    pt = orc.PTOracle(task, topo, train, test, R, mt, R * S, si, use_lg=lg, l_prob=0.5, lr=lr, seed=seed)
    w0 = np.stack([rep.w for rep in pt.replicas]).astype(np.float32)
    orc_c.adopt(pt, w0=w0.astype(np.float64))
    with np.errstate(all="ignore"):
        pt.run()
    s = parity.make_sampler(task, topo, train, test, R_local=R, R_global=R, first=0, S=S, si=si, use_lg=lg, lr=lr, seed=seed, schedule=sched)
    s.set_state(w0, np.array(pt.temperatures, dtype=np.float32))
    s.run(-1)
    s.sync()
    tr = s.traces()
    for r, rep in enumerate(pt.replicas):
        assert rep.num_accepted == S - 1                     # the oracle (pinned to the reference) accepts every NaN proposal ...
        np.testing.assert_array_equal(tr["accept"][r], np.arange(-1, S - 1).clip(0))     # ... and so does the kernel: 0, 0, 1, 2, ...
        assert np.isnan(tr["likeh"][r, 1:]).all() and np.isnan(rep.likeh[1:, 0]).all()
        if not lg:                                           # random walk: the weights stay finite and comparable
            np.testing.assert_allclose(tr["pos_w"][r], rep.pos_w, rtol=2e-5, atol=2e-5)
    assert (s.state()["num_accepted"] == S - 1).all()
    assert s.swap_stats()[1:] == (pt.total_swap_proposals, pt.rounds_done)
    assert [list(x) for x in s.swap_log()] == [list(x) for x in pt.src_log]      # NaN scalars: min(709, nan) == 709 decides (F5)
    s.close()
