"""Pins oracle/ptnn_oracle.py against vectors produced by the reference itself
(tests/golden/make_fixtures.py imported REG/CLS in the survey container).  CPU only."""
import glob
import json
import os

import numpy as np
import pytest

import ptnn_oracle as orc

RTOL = 1e-11


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name), allow_pickle=False))


FUNC_CASES = ["reg_sunspot_4_5_1", "reg_mackey_4_10_1", "cls_iris_4_12_3", "cls_ions_34_50_2", "reg_synth_32_96_1",
              "cls_ions_34_100_2", "reg_sunspot5_5_5_1", "reg_mackey5_5_10_1"]      # the last two: BASELINE configs 1 / 3 literally
DATA_OF = {"reg_sunspot5_5_5_1": "sunspot5", "reg_mackey5_5_10_1": "mackey5",
           "reg_sunspot_4_5_1": "sunspot", "reg_mackey_4_10_1": "mackey", "cls_iris_4_12_3": "iris",
           "cls_ions_34_50_2": "ions", "reg_synth_32_96_1": "synth32", "cls_ions_34_100_2": "ions"}


def test_philox_known_answer():
    # Random123 kat_vectors: philox4x32-10, counter 0/key 0 and the "pi" vector
    x = orc.philox4x32(0, 0, 0, 0, 0)
    assert [int(v) for v in x] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    seed = (0xa4093822 | (0x299f31d0 << 32))
    x = orc.philox4x32(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, seed)
    assert [int(v) for v in x] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("key", FUNC_CASES)
def test_forward_sgd_likelihood_prior(golden_dir, datasets, key):
    g = _load(golden_dir, f"functions_{key}.npz")
    topo = tuple(int(v) for v in g["topology"])
    task = int(g["task"])
    train, test = datasets[DATA_OF[key] + "_train"], datasets[DATA_OF[key] + "_test"]
    I = topo[0]
    for wi in range(3):
        w = g[f"w{wi}"]
        hid, out = orc.forward(train[:, :I], w, topo)
        hid_f, out_f = orc.forward(train[:, :I], w, topo, faithful=True)
        np.testing.assert_allclose(hid_f, hid, rtol=1e-13)
        np.testing.assert_allclose(hid[3], g[f"hid_row3_{wi}"], rtol=RTOL)
        np.testing.assert_allclose(out[3], g[f"out_row3_{wi}"], rtol=RTOL)
        if task == orc.TASK_REG:
            np.testing.assert_allclose(out[:, 0], g[f"fx{wi}"], rtol=RTOL)
        else:
            np.testing.assert_array_equal(np.argmax(out, axis=1), g[f"fx{wi}"])
            e = np.exp(out)
            np.testing.assert_allclose(e / e.sum(1, keepdims=True), g[f"prob{wi}"], rtol=RTOL)
        for lr in (0.1, 0.01):
            np.testing.assert_allclose(orc.langevin_gradient(train, w, topo, lr, task), g[f"lg{wi}_lr{lr}"],
                                       rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(orc.langevin_gradient(train[:1], w, topo, lr, task),
                                       g[f"lg1row{wi}_lr{lr}"], rtol=RTOL, atol=1e-14)
            perm = orc.langevin_gradient(train[::-1], w, topo, lr, task)
            np.testing.assert_allclose(perm, g[f"lgperm{wi}_lr{lr}"], rtol=1e-9, atol=1e-12)
            assert not np.allclose(perm, g[f"lg{wi}_lr{lr}"], rtol=1e-9, atol=1e-12)   # order matters (Q4)
        for T in (1.0, 1.2599210498948732, 2.0, 10.0):
            if task == orc.TASK_REG:
                for tau in (0.01, 0.1):
                    l, fx, r = orc.likelihood_reg(train, w, tau, topo, T)
                    np.testing.assert_allclose([l, r], g[f"lik{wi}_T{T}_tau{tau}"], rtol=RTOL)
                    l, fx, r = orc.likelihood_reg(test, w, tau, topo, T)
                    np.testing.assert_allclose([l, r], g[f"liktest{wi}_T{T}_tau{tau}"], rtol=RTOL)
                    np.testing.assert_allclose(orc.prior_reg(25, 0, 0, w, tau, topo), g[f"prior{wi}_tau{tau}"], rtol=RTOL)
            else:
                l, fx, r = orc.likelihood_cls(train, w, topo, T)
                np.testing.assert_allclose([l, r, orc.accuracy(fx, train[:, I])], g[f"lik{wi}_T{T}"], rtol=RTOL)
                l, fx, r = orc.likelihood_cls(test, w, topo, T)
                np.testing.assert_allclose([l, r, orc.accuracy(fx, test[:, I])], g[f"liktest{wi}_T{T}"], rtol=RTOL)
                np.testing.assert_allclose(orc.prior_cls(25, w, topo), g[f"prior{wi}"], rtol=RTOL)


def test_ladder(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "ladder.json")))
    for c in cases:
        T = orc.temperature_ladder(c["R"], c["Tmax"])
        assert [float(t).hex() for t in T] == c["T"]
        assert [str(t) for t in T] == c["s"]


def test_swap_cascade(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "swap_cascade.json")))
    assert len(cases) >= 28
    for c in cases:
        L = [float(x) for x in c["L"]]
        src, ns = orc.swap_cascade(L, c["u"])
        assert src == c["src"], c["tag"]
        assert ns == c["num_swap"], c["tag"]
        assert len(L) - 1 == c["total"]
        assert sorted(src) == list(range(len(L)))           # always a permutation


TRAJ = sorted(os.path.basename(p)[len("trajectory_"):-4]
              for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "trajectory_*.npz")))


@pytest.mark.parametrize("key", TRAJ)
def test_single_replica_trajectory(golden_dir, datasets, key):
    g = _load(golden_dir, f"trajectory_{key}.npz")
    topo = tuple(int(v) for v in g["topology"])
    task, S = int(g["task"]), int(g["S"])
    dname = str(g["dataset"])
    train, test = datasets[dname + "_train"], datasets[dname + "_test"]
    tape = orc.PhiloxTape(int(g["seed"]))
    rep = orc.Replica(task, topo, train, test, g["w0"], float(g["T"]), S, bool(g["use_lg"]), 0.5, float(g["lr"]),
                      tape, int(g["gid"]))
    for i in range(S - 1):
        rep.step(i)
    np.testing.assert_array_equal(rep.accept_list, g["accept_list"])          # every MH decision identical
    np.testing.assert_allclose(rep.pos_w, g["pos_w"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(rep.likeh, g["likeh"], rtol=1e-9, atol=1e-9)
    for nm in ("rmse_train", "rmse_test", "acc_train", "acc_test"):
        np.testing.assert_allclose(getattr(rep, nm), g[nm], rtol=1e-9, atol=1e-12)
    fin = g["final_param"]
    P = rep.P
    np.testing.assert_allclose(rep.w, fin[:P], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(rep.eta, fin[P], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(rep.likelihood, fin[P + 1], rtol=1e-9)
    assert fin[P + 2] == (rep.adapttemp if task == orc.TASK_REG else rep.T)
    np.testing.assert_allclose(100.0 * rep.num_accepted / S, g["accept_ratio"])
    if key.endswith("noswitch"):
        assert rep.adapttemp == rep.T                                         # Q9 float-equality trigger never fired
    else:
        assert rep.adapttemp == 1


SWAPTRAJ = ["reg", "reg_nophantom", "cls", "cls_nophantom", "reg_sunspot5"]


@pytest.mark.parametrize("key", SWAPTRAJ)
def test_full_pt_run_with_swaps(golden_dir, datasets, key):
    g = _load(golden_dir, f"swap_trajectory_{key}.npz")
    topo = tuple(int(v) for v in g["topology"])
    task = int(g["task"])
    dname = str(g["dataset"])
    R = int(g["R"])
    pt = orc.PTOracle(task, topo, datasets[dname + "_train"], datasets[dname + "_test"], R, int(g["maxtemp"]),
                      int(g["NumSample"]), int(g["si"]), use_lg=bool(g["use_lg"]), l_prob=0.5, lr=float(g["lr"]),
                      seed=int(g["seed"]))
    assert pt.temperatures == list(g["temperatures"])
    pt.run()
    assert pt.num_swap == int(g["num_swap"])
    assert pt.total_swap_proposals == int(g["total_swap_proposals"])
    assert pt.swap_perc == pytest.approx(float(g["swap_perc"]))
    for r, rep in enumerate(pt.replicas):
        np.testing.assert_array_equal(rep.accept_list, g[f"accept_list_{r}"])
        np.testing.assert_allclose(rep.pos_w, g[f"pos_w_{r}"], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(rep.likeh, g[f"likeh_{r}"], rtol=1e-9, atol=1e-9)
        for nm in ("rmse_train", "rmse_test", "acc_train", "acc_test"):
            np.testing.assert_allclose(getattr(rep, nm), g[f"{nm}_{r}"], rtol=1e-9, atol=1e-12)
    # phantom round accounting (Q13)
    S, si = pt.S, pt.si
    assert pt.rounds_done == int(S / si)
    assert (pt.rounds_done - orc.count_handoffs(task, S, si)) == (0 if "nophantom" in key else 1)
