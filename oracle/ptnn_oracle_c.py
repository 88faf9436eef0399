"""ctypes face of oracle/ptnn_oracle_c.c: the float64 C restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY (see the header of ptnn_oracle_c.c): used by tests/ to follow the device through whole runs of the
reference's standard length, where the numpy oracle (ptnn_oracle.py) would need the better part of an hour.  `CReplica` offers
the interface of `ptnn_oracle.Replica` (same attribute names, same trace arrays), so `PTOracle` can drive either.

    pt = ptnn_oracle.PTOracle(...)
    ptnn_oracle_c.adopt(pt)            # replaces pt.replicas by C-backed chains in the same state
"""
import ctypes as C
import os
import subprocess

import numpy as np

import ptnn_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ptnn_oracle_c.c")
LIB = os.path.join(HERE, "libptnn_oracle.so")


def build(force=False):
    """gcc -O2 -shared: seconds.  Rebuilt when the source is newer than the library."""
    if force or not os.path.exists(LIB) or os.path.getmtime(SRC) > os.path.getmtime(LIB):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"])
    return LIB


_dp = C.POINTER(C.c_double)


class _Rep(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ("task", "I", "H", "O", "P", "Ntr", "Nte", "ncols", "S", "use_lg", "gid", "noise_gid",
                                          "num_accepted", "langevin_count", "init_count", "lik_stale",
                                          "last_stale", "last_natural", "last_forced", "last_lg")] +
                [("seed", C.c_uint64)] +
                [(n, C.c_double) for n in ("T", "adapttemp", "l_prob", "lr", "step_w", "step_eta", "sigma_sq", "nu1", "nu2", "pt_samples",
                                           "eta", "tau_pro", "likelihood", "prior_current", "last_logalpha", "last_u", "last_scale",
                                           "last_lik_cur", "last_prior_cur")] +
                [(n, _dp) for n in ("train", "test", "w", "pos_w", "likeh", "accept_list", "rmse_train", "rmse_test", "acc_train",
                                    "acc_test", "scratch")])


_lib = None


def lib():
    global _lib
    if _lib is None:
        l_ = C.CDLL(build())
        assert l_.orc_replica_struct_bytes() == C.sizeof(_Rep), "ptnn_oracle_c.c and its ctypes mirror are out of step"
        l_.orc_replica_step.restype = C.c_int
        l_.orc_replica_step.argtypes = [C.POINTER(_Rep), C.c_int, C.c_int]
        l_.orc_replica_init.argtypes = [C.POINTER(_Rep)]
        l_.orc_replica_run.argtypes = [C.POINTER(_Rep), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        l_.orc_replica_set_state.argtypes = [C.POINTER(_Rep), _dp, C.c_double]
        l_.orc_likelihood.restype = C.c_double
        l_.orc_likelihood.argtypes = [C.c_int] * 4 + [_dp, C.c_int, C.c_int, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _dp]
        l_.orc_prior.restype = C.c_double
        l_.orc_prior.argtypes = [C.c_int] * 4 + [C.c_double] * 3 + [_dp, C.c_double]
        l_.orc_langevin_gradient.argtypes = [C.c_int] * 4 + [_dp, C.c_int, C.c_int, _dp, C.c_double, _dp, _dp]
        l_.orc_step_scalars.argtypes = [C.c_uint64, C.c_int, C.c_int, _dp]
        l_.orc_w_noise.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, _dp]
        l_.orc_w_init.argtypes = [C.c_uint64, C.c_int, C.c_int, _dp]
        _lib = l_
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- the pure functions (golden-vector tests) ----
def likelihood(task, topo, data, w, tau_sq, adapttemp):
    """-> (tempered loglik, fx[N], rmse, accuracy)"""
    data, w = _f64(data), _f64(w)
    fx = np.empty(data.shape[0])
    rm, ac = C.c_double(), C.c_double()
    sc = np.empty(topo[1] + topo[2])
    v = lib().orc_likelihood(task, *topo, _p(data), data.shape[0], data.shape[1], _p(w), float(tau_sq), float(adapttemp),
                             C.byref(rm), C.byref(ac), _p(fx), _p(sc))
    return v, fx, rm.value, ac.value


def prior(task, topo, w, tau_sq=1.0, sigma_squared=25.0, nu_1=0.0, nu_2=0.0):
    w = _f64(w)
    return lib().orc_prior(task, *topo, sigma_squared, nu_1, nu_2, _p(w), float(tau_sq))


def langevin_gradient(data, w, topo, lr, task):
    data, w = _f64(data), _f64(w)
    out = np.empty_like(w)
    sc = np.empty(2 * topo[1] + 2 * topo[2])
    lib().orc_langevin_gradient(task, *topo, _p(data), data.shape[0], data.shape[1], _p(w), float(lr), _p(out), _p(sc))
    return out


def step_scalars(seed, replica, step):
    out = np.empty(3)
    lib().orc_step_scalars(int(seed), replica, step, _p(out))
    return tuple(out)


def w_noise(seed, replica, step, n):
    out = np.empty(n)
    lib().orc_w_noise(int(seed), replica, step, n, _p(out))
    return out


class CReplica:
    """One chain in C (orc_replica), with the attribute names of ptnn_oracle.Replica."""

    def __init__(self, task, topo, train, test, w0, temperature, samples, use_lg, l_prob, lr, seed, gid, noise_gid=None,
                 step_w=0.025, step_eta=0.2, sigma_squared=25.0, nu_1=0.0, nu_2=0.0):
        self.task, self.topo = task, tuple(topo)
        I, H, O = self.topo
        P = orc.num_param(topo)
        self.P, self.S = P, int(samples)
        S = self.S
        self.train, self.test = _f64(train), _f64(test)
        assert self.train.shape[1] == self.test.shape[1]
        self._w = _f64(w0).copy()
        self.pos_w = np.ones((S, P))
        self.likeh = np.zeros((S, 2))
        self.likeh[0, :] = [-100, -100]
        self.accept_list = np.zeros(S)
        self.rmse_train, self.rmse_test = np.zeros(S), np.zeros(S)
        self.acc_train, self.acc_test = np.zeros(S), np.zeros(S)
        self._scratch = np.zeros(4 * P + 2 * H + 2 * O)
        self.gid = int(gid)
        c = _Rep()
        c.task, c.I, c.H, c.O, c.P = task, I, H, O, P
        c.Ntr, c.Nte, c.ncols, c.S = self.train.shape[0], self.test.shape[0], self.train.shape[1], S
        c.use_lg, c.gid, c.noise_gid = int(bool(use_lg)), int(gid), int(gid if noise_gid is None else noise_gid)
        c.seed = int(seed)
        c.T, c.l_prob, c.lr = float(temperature), float(l_prob), float(lr)
        c.step_w, c.step_eta, c.sigma_sq, c.nu1, c.nu2 = step_w, step_eta, sigma_squared, nu_1, nu_2
        c.pt_samples = S * 0.6
        c.train, c.test, c.w = _p(self.train), _p(self.test), _p(self._w)
        c.pos_w, c.likeh, c.accept_list = _p(self.pos_w), _p(self.likeh), _p(self.accept_list)
        c.rmse_train, c.rmse_test, c.acc_train, c.acc_test = _p(self.rmse_train), _p(self.rmse_test), _p(self.acc_train), _p(self.acc_test)
        c.scratch = _p(self._scratch)
        self.c = c
        lib().orc_replica_init(C.byref(c))

    # scalar state lives in the C struct
    def __getattr__(self, name):
        if name in ("eta", "tau_pro", "likelihood", "prior_current", "adapttemp", "T", "num_accepted", "langevin_count", "init_count",
                    "last_logalpha", "last_u", "last_scale", "noise_gid"):
            return getattr(self.__dict__["c"], name)
        if name in ("lik_stale", "last_stale", "last_natural", "last_forced"):
            return bool(getattr(self.__dict__["c"], name))
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in ("eta", "likelihood", "prior_current", "adapttemp", "T", "noise_gid"):
            setattr(self.c, name, value)
        elif name == "lik_stale":
            self.c.lik_stale = int(bool(value))
        elif name == "w":
            self._w[:] = value                               # a swap hands over VALUES; the C side keeps its buffer
        else:
            object.__setattr__(self, name, value)

    @property
    def w(self):
        return self._w.copy()

    def step(self, i, force=None):
        return bool(lib().orc_replica_step(C.byref(self.c), int(i), -1 if force is None else int(bool(force))))

    def run(self, i0, i1, force=None, sync_w=None, sync_eta=None):
        """Steps [i0, i1); force: None or int8 [i1 - i0] of -1 / 0 / 1.  sync_w: float32 [i1 - i0, >= P] (row k = the pos_w row
        another implementation recorded for MH step i0 + k), sync_eta: float32 [i1 - i0] (its eta after that step) or None:
        after every accepted step the chain continues from that state, likelihood and prior re-evaluated in float64
        (orc_replica_set_state).  -> dict of per-step records."""
        n = i1 - i0
        rec = dict(logalpha=np.empty(n), logu=np.empty(n), scale=np.empty(n), stale=np.empty(n, dtype=np.int8), natural=np.empty(n, dtype=np.int8))
        f = None if force is None else np.ascontiguousarray(force, dtype=np.int8)
        sw, ss, se, es = None, 0, None, 0
        if sync_w is not None:
            assert sync_w.dtype == np.float32 and sync_w.ndim == 2 and sync_w.shape[0] == n and sync_w.shape[1] >= self.P
            assert sync_w.strides[1] == 4 and sync_w.strides[0] % 4 == 0
            sw, ss = sync_w.ctypes.data, sync_w.strides[0] // 4
            if sync_eta is not None:
                assert sync_eta.dtype == np.float32 and sync_eta.shape == (n,) and sync_eta.strides[0] % 4 == 0
                se, es = sync_eta.ctypes.data, sync_eta.strides[0] // 4
        extra = [None, None, None]
        if sync_w is not None:
            # the accepted steps re-evaluated at the other side's own proposal (NaN on rejected steps): see orc_replica_run
            rec.update(la_sync=np.empty(n), scale_sync=np.empty(n), lik_sync=np.empty(n))
            extra = [rec[k].ctypes.data for k in ("la_sync", "scale_sync", "lik_sync")]
        lib().orc_replica_run(C.byref(self.c), int(i0), int(i1), None if f is None else f.ctypes.data, rec["logalpha"].ctypes.data,
                              rec["logu"].ctypes.data, rec["scale"].ctypes.data, rec["stale"].ctypes.data, rec["natural"].ctypes.data,
                              sw, ss, se, es, *extra)
        return rec

    def set_state(self, w, eta=None):
        """Continue from another implementation's state: (w, eta) replaced, the cached likelihood / prior re-evaluated from it in
        float64 at the chain's present temperature (orc_replica_set_state).  eta None (or classification): keep the chain's own."""
        w = _f64(w)
        assert w.shape == (self.P,)
        lib().orc_replica_set_state(C.byref(self.c), _p(w), float("nan") if eta is None else float(eta))

    def posted_L(self):
        """Q11: REG posts likelihood * T (REG:430), CLS the tempered likelihood (CLS:439)."""
        return self.likelihood * self.T if self.task == orc.TASK_REG else self.likelihood


def adopt(pt, w0=None):
    """Replace the (not yet advanced) numpy chains of a PTOracle by C chains from the same initial state.  w0: [R, P] initial
    weights (default: the chains' own)."""
    reps = []
    for r, rep in enumerate(pt.replicas):
        assert rep.num_accepted == 0 and not rep.accept_list.any(), "adopt() wants chains that have not run yet"
        w = rep.w if w0 is None else w0[r]
        reps.append(CReplica(pt.task, pt.topo, pt.train, pt.test, w, rep.T, rep.S, rep.use_lg, rep.l_prob, rep.lr, pt.tape.seed, rep.gid,
                             noise_gid=rep.noise_gid, step_w=rep.step_w, step_eta=rep.step_eta, sigma_squared=rep.sigma_squared,
                             nu_1=rep.nu_1, nu_2=rep.nu_2))
    pt.replicas = reps
    pt.holder_hist = [(0, list(reps))]
    return pt
