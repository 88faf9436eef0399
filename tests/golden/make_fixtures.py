#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by IMPORTING the reference (survey container only).

Runs only where /root/reference exists.  Nothing from the reference's source is copied: the script
imports `pt_timeseries_regression` (REG) and `pt_classification` (CLS) from where they lie,
replaces numpy's / `random`'s global draws with the Philox tape specified in
oracle/ptnn_oracle.py (so that every consumer reads the same variates), calls the reference's
own functions and stores inputs + outputs as small .npz / .json files.

    python tests/golden/make_fixtures.py            # F1..F8 (about a minute)
    python tests/golden/make_fixtures.py --stats    # F9 long statistical runs (tens of minutes)

Fixture ids follow SURVEY.md section 8c.
"""
import argparse
import contextlib
import importlib.util
import io
import json
import multiprocessing
import os
import random as pyrandom
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ptnn_oracle as orc  # noqa: E402  (tape specification only)

REF = "/root/reference"
REG_PATH = os.path.join(REF, "multicore-pt-regression", "pt_timeseries_regression.py")
CLS_PATH = os.path.join(REF, "multicore-pt-classification", "pt_classification.py")


def _import(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        spec.loader.exec_module(mod)
    return mod


REG = _import("ref_reg", REG_PATH)
CLS = _import("ref_cls", CLS_PATH)


# --------------------------------------------------------------------------------------------
# datasets (data, not code): re-emitted so tests and bench can run where /root/reference is absent
# --------------------------------------------------------------------------------------------
def load_datasets():
    d = {}
    base = os.path.join(REF, "multicore-pt-regression", "Data_OneStepAhead")
    for name in ("Sunspot", "Mackey", "Lazer"):
        d[name.lower() + "_train"] = np.loadtxt(os.path.join(base, name, "train.txt"))
        d[name.lower() + "_test"] = np.loadtxt(os.path.join(base, name, "test.txt"))
        d[name.lower() + "_scaled"] = np.loadtxt(os.path.join(base, name, "scaled_dataset.txt")).reshape(-1)
    cbase = os.path.join(REF, "multicore-pt-classification", "DATA")
    iris = np.genfromtxt(os.path.join(cbase, "iris.csv"), delimiter=";")
    classes = iris[:, 4].reshape(-1, 1) - 1          # CLS:921
    feats = iris[:, 0:4].copy()
    for k in range(4):                               # CLS:1003-1007 z-score
        feats[:, k] = (feats[:, k] - np.mean(feats[:, k])) / np.std(feats[:, k])
    idx = np.random.default_rng(2024).permutation(150)   # reference uses an unseeded permutation (CLS:1010)
    ntr = int(0.7 * 150)
    d["iris_train"] = np.hstack([feats[idx[:ntr]], classes[idx[:ntr]]])
    d["iris_test"] = np.hstack([feats[idx[ntr:]], classes[idx[ntr:]]])
    d["ions_train"] = np.genfromtxt(os.path.join(cbase, "Ions", "Ions", "ftrain.csv"), delimiter=",")[:, :-1]
    d["ions_test"] = np.genfromtxt(os.path.join(cbase, "Ions", "Ions", "ftest.csv"), delimiter=",")[:, :-1]
    d["cancer_train"] = np.genfromtxt(os.path.join(cbase, "Cancer", "ftrain.txt"), delimiter=" ")[:, :-1]
    d["cancer_test"] = np.genfromtxt(os.path.join(cbase, "Cancer", "ftest.txt"), delimiter=" ")[:, :-1]
    # BASELINE configs 1 and 3 name FIVE-input nets ("FNN 5-5-1", "FNN 5-10-1"); the shipped embedding has 4 lag inputs (SURVEY 8:
    # "Input Neuron: 4").  The literal variant: the same scaled series re-embedded with window 6 (5 lags + target), stride 2,
    # split like the shipped files (first int(0.6 n) rows train, last int(0.4 n) - 1 rows test)
    for name in ("sunspot", "mackey"):
        d[name + "5_train"], d[name + "5_test"] = reembed(d[name + "_scaled"], window=6, stride=2)
    # small sibling of BASELINE config 5: teacher FNN on uniform inputs, 32 features, targets in [0,1]
    d["synth32_train"], d["synth32_test"] = synthetic_regression(96, 64, 32, 96, seed=5)
    return d


def reembed(series, window, stride):
    """Rows series[stride k : stride k + window]; the shipped train/test files are this with window 5 (checked in
    tests/test_host_cpu.py against the shipped Sunspot / Mackey / Lazer files)."""
    series = np.asarray(series, dtype=np.float64).reshape(-1)
    n = (series.shape[0] - window) // stride + 1
    rows = np.stack([series[stride * k:stride * k + window] for k in range(n)])
    return rows[:int(0.6 * n)], rows[n - (int(0.4 * n) - 1):]


def synthetic_regression(n_rows, n_train, n_in, n_hidden, seed):
    """SURVEY.md 8(d) config 5 recipe: X ~ U(0,1), teacher FNN [n_in, n_hidden, 1] with N(0,1)/sqrt(fan_in) weights run
    through the reference forward rule (bias subtracted, sigmoid output), y = clip(teacher + N(0, 0.02^2), 0, 1)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (n_rows, n_in))
    topo = (n_in, n_hidden, 1)
    w_t = np.concatenate([rng.standard_normal(n_in * n_hidden) / np.sqrt(n_in), rng.standard_normal(n_hidden) / np.sqrt(n_hidden),
                          rng.standard_normal(n_hidden) / np.sqrt(n_in), rng.standard_normal(1) / np.sqrt(n_hidden)])
    y = np.clip(orc.forward(X, w_t, topo)[1][:, 0] + rng.normal(0, 0.02, n_rows), 0, 1)
    data = np.hstack([X, y[:, None]])
    return data[:n_train], data[n_train:]


# --------------------------------------------------------------------------------------------
# random tape patch
# --------------------------------------------------------------------------------------------
class RefTape:
    """Replaces np.random.{uniform,normal,randn} and random.uniform while the reference runs."""

    def __init__(self, seed, num_chains=1):
        self.tape = orc.PhiloxTape(seed)
        self.R = num_chains
        self.replica = None      # None = parent process
        self.step = -1
        self.init_idx = 0
        self.swap_round = 0
        self.swap_k = 0
        self.scripted_swap_u = None

    # np.random.uniform(0,1,1) in a replica = lx and starts a new step; np.random.uniform(0,1) in the parent = swap u
    def uniform(self, low=0.0, high=1.0, size=None):
        if size is None:
            if self.scripted_swap_u is not None:
                u = self.scripted_swap_u[self.swap_k]
                self.swap_k += 1
                return u
            u = self.tape.swap_uniforms(self.swap_round, self.R - 1)[self.swap_k]
            self.swap_k += 1
            if self.swap_k == self.R - 1:
                self.swap_k = 0
                self.swap_round += 1
            return float(u)
        self.step += 1
        self._scal = self.tape.step_scalars(self.replica, self.step)
        return np.array([self._scal[0]])

    def normal(self, loc=0.0, scale=1.0, size=None):
        if size == 1:                                   # eta proposal noise (REG:355)
            return np.array([loc + scale * self._scal[2]])
        n = self.tape.w_noise(self.replica, self.step, int(size))
        return loc + scale * n

    def randn(self, *shape):
        if self.replica is None and len(shape) == 1:     # parent: w0 of the next chain (REG:649)
            w = self.tape.w_init(self.init_idx, shape[0])
            self.init_idx += 1
            return w
        return np.zeros(shape)                           # discarded draws (REG:256, Network.__init__)

    def py_uniform(self, a, b):
        return self._scal[1]


@contextlib.contextmanager
def patched(tape, capture=None):
    saved = (np.random.uniform, np.random.normal, np.random.randn, pyrandom.uniform, np.savetxt)
    np.random.uniform, np.random.normal, np.random.randn = tape.uniform, tape.normal, tape.randn
    pyrandom.uniform = tape.py_uniform
    orig_savetxt = saved[4]

    def savetxt(fname, X, *a, **k):
        orig_savetxt(fname, X, *a, **k)
        np.save(str(fname) + ".npy", np.asarray(X, dtype=np.float64))     # full precision twin
        if capture is not None:
            capture[os.path.basename(os.path.dirname(str(fname))) + "/" + os.path.basename(str(fname))] = \
                np.array(X, dtype=np.float64, copy=True)

    np.savetxt = savetxt
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            yield
    finally:
        np.random.uniform, np.random.normal, np.random.randn, pyrandom.uniform, np.savetxt = saved


class FakeQueue:
    def __init__(self, items=()):
        self.items = list(items)

    def put(self, x):
        self.items.append(x)

    def get(self):
        return self.items.pop(0)


class FakeEvent:
    def set(self): pass
    def clear(self): pass
    def wait(self): return True


SUBDIRS = ["predictions", "posterior", "results", "surrogate", "surrogate/learnsurrogate_data",
           "posterior/pos_w", "posterior/pos_likelihood", "posterior/surg_likelihood", "posterior/accept_list"]


def mkdirs(path):
    for s in SUBDIRS:
        os.makedirs(os.path.join(path, s), exist_ok=True)


def seeded_w(seed, P):
    return orc.PhiloxTape(seed).w_init(7, P)


# --------------------------------------------------------------------------------------------
# F1..F3: pure functions
# --------------------------------------------------------------------------------------------
CASES = [  # (key, module, task, topology, dataset)
    ("reg_sunspot_4_5_1", "REG", orc.TASK_REG, [4, 5, 1], "sunspot"),
    ("reg_mackey_4_10_1", "REG", orc.TASK_REG, [4, 10, 1], "mackey"),
    ("cls_iris_4_12_3", "CLS", orc.TASK_CLS, [4, 12, 3], "iris"),
    ("cls_ions_34_50_2", "CLS", orc.TASK_CLS, [34, 50, 2], "ions"),
    ("reg_synth_32_96_1", "REG", orc.TASK_REG, [32, 96, 1], "synth32"),      # H > 64: the multi-wave (wide) kernels
    ("cls_ions_34_100_2", "CLS", orc.TASK_CLS, [34, 100, 2], "ions"),
    ("reg_sunspot5_5_5_1", "REG", orc.TASK_REG, [5, 5, 1], "sunspot5"),       # BASELINE config 1's literal topology
    ("reg_mackey5_5_10_1", "REG", orc.TASK_REG, [5, 10, 1], "mackey5"),       # BASELINE config 3's literal topology
]


def make_replica(modname, topo, train, test, T, S, use_lg, lr, l_prob, si, path, q=None):
    q = q or FakeQueue()
    if modname == "REG":
        return REG.ptReplica(use_lg, lr, None, None, None, S, train, test, topo, 0.5, T, si, l_prob, path, q,
                             FakeEvent(), FakeEvent())
    return CLS.ptReplica(use_lg, lr, None, None, None, S, train, test, topo, 0.5, T, si, path, q,
                         FakeEvent(), FakeEvent())


def gen_functions(ds, out, only=None):
    for key, modname, task, topo, dname in CASES:
        if only and "functions_" + key not in only:
            continue
        mod = REG if modname == "REG" else CLS
        train, test = ds[dname + "_train"], ds[dname + "_test"]
        P = orc.num_param(topo)
        rec = {"topology": np.array(topo), "task": np.array(task)}
        for wi, w in enumerate([np.linspace(-1, 1, P), seeded_w(11, P), 0.3 * seeded_w(12, P)]):
            with contextlib.redirect_stdout(io.StringIO()):
                net = mod.Network(topo, train, test, 0.1)
                res = net.evaluate_proposal(train, w.copy())
                # hidden activations of the last row, to pin R2 itself
                net.decode(w.copy())
                net.ForwardPass(train[3, :topo[0]])
            rec[f"w{wi}"] = w
            if task == orc.TASK_REG:
                rec[f"fx{wi}"] = np.asarray(res)
            else:
                rec[f"fx{wi}"] = np.asarray(res[0])
                rec[f"prob{wi}"] = np.asarray(res[1])
            rec[f"hid_row3_{wi}"] = np.asarray(net.hidout).reshape(-1)
            rec[f"out_row3_{wi}"] = np.asarray(net.out).reshape(-1)
            # F2: one SGD row and one epoch
            for lr in (0.1, 0.01):
                with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                    net = mod.Network(topo, train, test, lr)
                    rec[f"lg{wi}_lr{lr}"] = net.langevin_gradient(train, w.copy(), 1)
                    rec[f"lg1row{wi}_lr{lr}"] = net.langevin_gradient(train[:1], w.copy(), 1)
                    rec[f"lgperm{wi}_lr{lr}"] = net.langevin_gradient(train[::-1], w.copy(), 1)
            # F3: likelihood / prior
            tmp = tempfile.mkdtemp()
            for T in (1.0, 1.2599210498948732, 2.0, 10.0):
                rep = make_replica(modname, topo, train, test, T, 10, False, 0.1, 0.5, 100, tmp)
                rep.adapttemp = T
                with contextlib.redirect_stdout(io.StringIO()):
                    net = mod.Network(topo, train, test, 0.1)
                    if task == orc.TASK_REG:
                        for tau in (0.01, 0.1):
                            l, fx, r = rep.likelihood_func(net, train, w.copy(), tau)
                            rec[f"lik{wi}_T{T}_tau{tau}"] = np.array([l, r])
                            l, fx, r = rep.likelihood_func(net, test, w.copy(), tau)
                            rec[f"liktest{wi}_T{T}_tau{tau}"] = np.array([l, r])
                            rec[f"prior{wi}_tau{tau}"] = np.array(rep.prior_likelihood(25, 0, 0, w.copy(), tau))
                    else:
                        l, fx, r = rep.likelihood_func(net, train, w.copy())
                        rec[f"lik{wi}_T{T}"] = np.array([l, r, rep.accuracy(fx, train[:, topo[0]])])
                        l, fx, r = rep.likelihood_func(net, test, w.copy())
                        rec[f"liktest{wi}_T{T}"] = np.array([l, r, rep.accuracy(fx, test[:, topo[0]])])
                        rec[f"prior{wi}"] = np.array(rep.prior_likelihood(25, 0, 0, w.copy()))
            shutil.rmtree(tmp)
        np.savez_compressed(os.path.join(out, f"functions_{key}.npz"), **rec)
        print("F1-F3", key)


# --------------------------------------------------------------------------------------------
# F4: single-replica trajectories (in-process ptReplica.run with the tape)
# --------------------------------------------------------------------------------------------
TRAJ = [  # key, module, task, topo, dataset, use_lg, lr, T, S, seed
    ("reg_rw", "REG", orc.TASK_REG, [4, 5, 1], "sunspot", False, 0.1, 1.2599210498948732, 200, 101),
    ("reg_lg", "REG", orc.TASK_REG, [4, 5, 1], "sunspot", True, 0.1, 1.5874010519681994, 100, 102),
    ("reg_lg_mackey", "REG", orc.TASK_REG, [4, 10, 1], "mackey", True, 0.1, 1.0, 60, 103),
    ("cls_rw", "CLS", orc.TASK_CLS, [4, 12, 3], "iris", False, 0.01, 2.154434690031884, 200, 104),
    ("cls_lg", "CLS", orc.TASK_CLS, [4, 12, 3], "iris", True, 0.01, 1.0, 100, 105),
    ("cls_rw_ions", "CLS", orc.TASK_CLS, [34, 50, 2], "ions", False, 0.01, 1.6681005372000588, 60, 106),
    ("reg_rw_noswitch", "REG", orc.TASK_REG, [4, 5, 1], "sunspot", False, 0.1, 2.0, 57, 107),  # 0.6*57 not integral
    ("reg_lg_wide", "REG", orc.TASK_REG, [32, 96, 1], "synth32", True, 0.1, 1.2599210498948732, 30, 108),
    ("cls_lg_wide", "CLS", orc.TASK_CLS, [34, 100, 2], "ions", True, 0.01, 1.0, 30, 109),
    ("reg_lg_sunspot5", "REG", orc.TASK_REG, [5, 5, 1], "sunspot5", True, 0.1, 1.2599210498948732, 100, 110),
    ("reg_lg_mackey5", "REG", orc.TASK_REG, [5, 10, 1], "mackey5", True, 0.1, 1.5874010519681994, 60, 111),
]


def gen_trajectories(ds, out, only=None):
    for key, modname, task, topo, dname, use_lg, lr, T, S, seed in TRAJ:
        if only and "trajectory_" + key not in only:
            continue
        train, test = ds[dname + "_train"], ds[dname + "_test"]
        P = orc.num_param(topo)
        tape = RefTape(seed)
        tape.replica = 3                                  # arbitrary global replica id
        w0 = orc.PhiloxTape(seed).w_init(3, P)
        tmp = tempfile.mkdtemp()
        mkdirs(tmp)
        q = FakeQueue()
        rep = make_replica(modname, topo, train, test, T, S, use_lg, lr, 0.5, 10 * S, tmp, q)
        rep.w = w0.copy()
        cap = {}
        with patched(tape, cap), np.errstate(all="ignore"):
            rep.run()
        final = q.items[-1]
        rec = dict(topology=np.array(topo), task=np.array(task), use_lg=np.array(use_lg), lr=np.array(lr),
                   T=np.array(T), S=np.array(S), seed=np.array(seed), gid=np.array(3), w0=w0,
                   dataset=np.array(dname), final_param=np.asarray(final, dtype=np.float64))
        tn = str(T)
        rec["pos_w"] = cap[f"pos_w/chain_{tn}.txt"]
        rec["likeh"] = cap[f"pos_likelihood/chain_{tn}.txt"]
        rec["accept_list"] = cap[f"accept_list/chain_{tn}.txt"]
        rec["accept_ratio"] = cap[f"accept_list/chain_{tn}_accept.txt"]
        for nm in ("rmse_train", "rmse_test", "acc_train", "acc_test"):
            rec[nm] = cap[f"predictions/{nm}_chain_{tn}.txt"]
        np.savez_compressed(os.path.join(out, f"trajectory_{key}.npz"), **rec)
        shutil.rmtree(tmp)
        print("F4", key, "accepted", int(rec["accept_list"][-1]))


# --------------------------------------------------------------------------------------------
# F5: swap cascade driven through the reference's swap_procedure
# --------------------------------------------------------------------------------------------
def gen_cascade(out):
    rng = np.random.default_rng(55)
    cases = []
    P = 3

    def run_case(L, u):
        R = len(L)
        pt = REG.ParallelTempering(False, 0.1, np.zeros((2, 5)), np.zeros((2, 5)), [1, 1, 0], R, 2, 100, 10, 0.5, "/tmp")
        pt.num_param = P
        queues = [FakeQueue([np.array([k, k, k, 0.0, L[k], 1.0])]) for k in range(R)]
        tape = RefTape(0, R)
        tape.scripted_swap_u = list(u)
        with patched(tape), np.errstate(all="ignore"):
            for k in range(R - 1):
                p1, p2, sw = pt.swap_procedure(queues[k], queues[k + 1])
                queues[k].put(p1)
                queues[k + 1].put(p2)
        src = [int(q.items[0][0]) for q in queues]
        return src, int(pt.num_swap), int(pt.total_swap_proposals)

    def add(L, u, tag):
        src, ns, tot = run_case(L, u)
        cases.append(dict(tag=tag, L=[repr(float(x)) for x in L], u=[float(x) for x in u], src=src, num_swap=ns, total=tot))

    for R in (2, 4, 7, 64):
        L = rng.normal(0, 3, R)
        add(L, rng.uniform(0, 1, R - 1), f"random_R{R}")
        add(L, np.zeros(R - 1), f"allswap_R{R}")
        add(L, np.ones(R - 1) * 0.999999, f"highu_R{R}")
        add(np.sort(L)[::-1].copy(), rng.uniform(0, 1, R - 1), f"descending_R{R}")
        add(np.zeros(R), np.full(R - 1, 0.5), f"ties_u0.5_R{R}")
        add(np.zeros(R), np.full(R - 1, 0.4999), f"ties_u0.4999_R{R}")
    add([0.0, 800.0, -800.0, 5.0], [0.3, 0.3, 0.3], "big_diffs")
    add([float("inf"), 1.0, float("-inf"), float("inf")], [0.3, 0.3, 0.3], "infs")
    add([float("nan"), 1.0, 2.0, float("nan"), 0.0], [0.9, 0.9, 0.9, 0.9], "nans")
    add([1.0, float("nan"), 2.0], [0.99, 0.99], "nan_mid")
    with open(os.path.join(out, "swap_cascade.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print("F5", len(cases), "cases")


# --------------------------------------------------------------------------------------------
# F6: full multi-process run_chains with the tape (children inherit the patch through fork)
# --------------------------------------------------------------------------------------------
SWAPTRAJ = [  # key, module, task, topo, dataset, use_lg, lr, R, maxtemp, NumSample, si, seed
    ("reg", "REG", orc.TASK_REG, [4, 5, 1], "sunspot", True, 0.1, 4, 2, 400, 10, 201),      # S=100: S%si==0 -> phantom
    ("reg_nophantom", "REG", orc.TASK_REG, [4, 5, 1], "sunspot", False, 0.1, 4, 2, 412, 10, 202),  # S=103
    ("cls", "CLS", orc.TASK_CLS, [4, 12, 3], "iris", False, 0.01, 4, 10, 400, 10, 203),      # S=100 phantom
    ("cls_nophantom", "CLS", orc.TASK_CLS, [4, 12, 3], "iris", True, 0.01, 5, 10, 515, 10, 204),   # S=103
    # BASELINE config 1 as worded: Sunspot, FNN 5-5-1, 4 replicas, Langevin proposals, through the reference's own run_chains()
    ("reg_sunspot5", "REG", orc.TASK_REG, [5, 5, 1], "sunspot5", True, 0.1, 4, 2, 400, 10, 205),
]


def gen_swap_trajectories(ds, out, only=None):
    for key, modname, task, topo, dname, use_lg, lr, R, maxtemp, NumSample, si, seed in SWAPTRAJ:
        if only and "swap_trajectory_" + key not in only:
            continue
        mod = REG if modname == "REG" else CLS
        train, test = ds[dname + "_train"], ds[dname + "_test"]
        tmp = tempfile.mkdtemp()
        mkdirs(tmp)
        tape = RefTape(seed, R)
        orig_run = mod.ptReplica.run

        def run_wrapper(self, _orig=orig_run, _tape=tape):
            _tape.replica = self._gid                     # executes in the child after fork
            _tape.step = -1
            _orig(self)

        mod.ptReplica.run = run_wrapper
        try:
            with patched(tape), np.errstate(all="ignore"):
                if modname == "REG":
                    pt = mod.ParallelTempering(use_lg, lr, train, test, topo, R, maxtemp, NumSample, si, 0.5, tmp)
                else:
                    pt = mod.ParallelTempering(use_lg, lr, train, test, topo, R, maxtemp, NumSample, si, tmp)
                pt.initialize_chains(0.5)
                for g, ch in enumerate(pt.chains):
                    ch._gid = g
                res = pt.run_chains()
        finally:
            mod.ptReplica.run = orig_run
        S = pt.NumSamples
        rec = dict(topology=np.array(topo), task=np.array(task), use_lg=np.array(use_lg), lr=np.array(lr),
                   R=np.array(R), maxtemp=np.array(maxtemp), NumSample=np.array(NumSample), si=np.array(si),
                   seed=np.array(seed), dataset=np.array(dname), S=np.array(S),
                   temperatures=np.array(pt.temperatures), num_swap=np.array(pt.num_swap),
                   total_swap_proposals=np.array(pt.total_swap_proposals), swap_perc=np.array(res[8]))
        names = ["pos_w", "fx_train", "fx_test", "rmse_train", "rmse_test", "acc_train", "acc_test",
                 "likelihood_vec", "swap_perc", "accept_vec", "accept"]
        for nm, val in zip(names, res):
            if nm in ("fx_train", "fx_test"):
                rec["ret_" + nm + "_shape"] = np.array(np.shape(val))
                assert not np.any(val)
            else:
                rec["ret_" + nm] = np.asarray(val)
        for g, T in enumerate(pt.temperatures):
            tn = str(T)
            rec[f"pos_w_{g}"] = np.load(os.path.join(tmp, "posterior/pos_w", f"chain_{tn}.txt.npy"))
            rec[f"likeh_{g}"] = np.load(os.path.join(tmp, "posterior/pos_likelihood", f"chain_{tn}.txt.npy"))
            rec[f"accept_list_{g}"] = np.load(os.path.join(tmp, "posterior/accept_list", f"chain_{tn}.txt.npy"))
            for nm in ("rmse_train", "rmse_test", "acc_train", "acc_test"):
                rec[f"{nm}_{g}"] = np.load(os.path.join(tmp, "predictions", f"{nm}_chain_{tn}.txt.npy"))
        np.savez_compressed(os.path.join(out, f"swap_trajectory_{key}.npz"), **rec)
        if key == "reg" and not only:
            gen_layout(tmp, out)
        shutil.rmtree(tmp)
        print("F6", key, "swap_perc", float(res[8]), "num_swap", pt.num_swap, "/", pt.total_swap_proposals)


# F8: directory layout of one real run
def gen_layout(path, out):
    tree = {}
    for dp, dn, fn in os.walk(path):
        rel = os.path.relpath(dp, path)
        for d in dn:
            tree[os.path.normpath(os.path.join(rel, d)) + "/"] = None
        for f in fn:
            if f.endswith(".npy"):
                continue
            full = os.path.join(dp, f)
            with open(full) as fh:
                lines = fh.read().splitlines()
            tree[os.path.normpath(os.path.join(rel, f))] = dict(
                nlines=len(lines), ncols=len(lines[0].split()) if lines else 0,
                first=lines[0] if lines else "", last=lines[-1] if lines else "")
    with open(os.path.join(out, "layout_tree.json"), "w") as f:
        json.dump(tree, f, indent=0, sort_keys=True)
    print("F8 layout", len(tree), "entries")


# F7: ladder
def gen_ladder(out):
    cases = []
    for R, Tmax in [(4, 2), (10, 2), (10, 10), (16, 10), (64, 2), (256, 10), (1024, 10), (2, 2), (5, 10)]:
        pt = REG.ParallelTempering(False, 0.1, np.zeros((2, 5)), np.zeros((2, 5)), [4, 5, 1], R, Tmax, 100, 10, 0.5, "/tmp")
        with contextlib.redirect_stdout(io.StringIO()):
            pt.assign_temperatures()
        cases.append(dict(R=R, Tmax=Tmax, T=[float(t).hex() for t in pt.temperatures], s=[str(t) for t in pt.temperatures]))
    with open(os.path.join(out, "ladder.json"), "w") as f:
        json.dump(cases, f)
    print("F7 ladder")


# --------------------------------------------------------------------------------------------
# F9: statistical targets from long reference runs with the reference's OWN random numbers
# --------------------------------------------------------------------------------------------
# Every run of a configuration starts from the SAME initial weights (the parent's np.random is seeded with W0_SEED for
# initialize_chains, REG:649) and differs in the noise only (np.random / random re-seeded with `seed` before run_chains; the
# forked chains inherit that state, REG:709-712).  The spread of a statistic across the seeds is then the reference's own Monte
# Carlo error for that start, which is what the GPU runs from the same start are held against.
W0_SEED = 4242
STATS = [  # key, module, topo, dataset, use_lg, lr, R, maxtemp, S per replica, swap_ratio, seeds
    ("sunspot_rw_r8", "REG", [4, 5, 1], "sunspot", False, 0.1, 8, 2, 5000, 0.01, (1, 2, 3, 4, 5)),
    ("sunspot_lg_r8", "REG", [4, 5, 1], "sunspot", True, 0.1, 8, 2, 2500, 0.01, (1, 2, 3, 4, 5)),
    ("iris_rw_r8", "CLS", [4, 12, 3], "iris", False, 0.01, 8, 10, 5000, 0.02, (1, 2, 3, 4, 5)),
    ("mackey_lg_r8", "REG", [4, 10, 1], "mackey", True, 0.1, 8, 2, 2500, 0.01, (1, 2, 3, 4, 5)),
    ("ions_rw_r8", "CLS", [34, 50, 2], "ions", False, 0.01, 8, 10, 1500, 0.02, (1, 2, 3, 4, 5)),
    # the BASELINE metric's own shape (bench.py's default workload): 64 chains x 10 000 samples, swap interval 100.  64 forked
    # chains on 8 cores: about 16 min (Langevin) / 6 min (random walk) per run, so `--stats --only sunspot_lg_r64 sunspot_rw_r64`
    ("sunspot_lg_r64", "REG", [4, 5, 1], "sunspot", True, 0.1, 64, 2, 10000, 0.01, (1, 2, 3, 4, 5)),
    ("sunspot_rw_r64", "REG", [4, 5, 1], "sunspot", False, 0.1, 64, 2, 10000, 0.01, (1, 2, 3, 4, 5)),
]


def gen_stats(ds, out, only=None):
    for key, modname, topo, dname, use_lg, lr, R, maxtemp, S, ratio, seeds in STATS:
        if only and key not in only:
            continue
        mod = REG if modname == "REG" else CLS
        train, test = ds[dname + "_train"], ds[dname + "_test"]
        NumSample = S * R
        si = int(ratio * NumSample / R)
        runs = []
        for seed in seeds:
            tmp = tempfile.mkdtemp()
            mkdirs(tmp)
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                if modname == "REG":
                    pt = mod.ParallelTempering(use_lg, lr, train, test, topo, R, maxtemp, NumSample, si, 0.5, tmp)
                else:
                    pt = mod.ParallelTempering(use_lg, lr, train, test, topo, R, maxtemp, NumSample, si, tmp)
                np.random.seed(W0_SEED)
                pt.initialize_chains(0.5)
                w0 = [np.asarray(c.w, dtype=np.float64).tolist() for c in pt.chains]
                np.random.seed(seed)
                pyrandom.seed(seed)
                res = pt.run_chains()
            wall = time.time() - t0
            pos_w = res[0]                                   # (P, R*(S-b))
            b = int(S * 0.5)
            per = pos_w.reshape(pos_w.shape[0], R, S - b)
            acc_vec = res[9]
            runs.append(dict(seed=seed, wall_s=wall, samples_per_s=NumSample / wall, swap_perc=float(res[8]),
                             accept_pct=[float(100.0 * acc_vec[r, -1] / S) for r in range(R)],
                             w_mean=per.mean(axis=2).T.tolist(), w_var=per.var(axis=2).T.tolist(),
                             rmse_train_mean=float(res[3].mean()), rmse_test_mean=float(res[4].mean()),
                             acc_train_mean=float(res[5].mean()), acc_test_mean=float(res[6].mean())))
            shutil.rmtree(tmp)
            print("F9", key, "seed", seed, "%.1fs" % wall, "swap%%=%.2f" % float(res[8]), flush=True)
        with open(os.path.join(out, f"stats_{key}.json"), "w") as f:
            # scalars as JSON; the arrays (initial weights float64, per-run posterior means / variances float32) beside it
            np.savez_compressed(os.path.join(out, f"stats_{key}.npz"), w0=np.array(w0, dtype=np.float64),
                                w_mean=np.array([r.pop("w_mean") for r in runs], dtype=np.float32),
                                w_var=np.array([r.pop("w_var") for r in runs], dtype=np.float32))
            json.dump(dict(key=key, module=modname, topology=topo, dataset=dname, use_lg=use_lg, lr=lr, R=R,
                           maxtemp=maxtemp, S=S, swap_interval=si, cores=os.cpu_count(), w0_seed=W0_SEED, runs=runs,
                           arrays=f"stats_{key}.npz"), f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", action="store_true")
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    multiprocessing.set_start_method("fork", force=True)
    ds = load_datasets()
    if a.stats:
        gen_stats(ds, a.out, a.only)
        return
    np.savez_compressed(os.path.join(a.out, "datasets.npz"), **ds)
    if not a.only:                       # --only functions_<key> trajectory_<key> swap_trajectory_<key>: just those files (+ datasets.npz)
        gen_ladder(a.out)
        gen_cascade(a.out)
    gen_functions(ds, a.out, a.only)
    gen_trajectories(ds, a.out, a.only)
    gen_swap_trajectories(ds, a.out, a.only)


if __name__ == "__main__":
    main()
