"""CPU oracle: float64 numpy restatement of the reference's parallel-tempering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (`parallel-tempering-neural-net_amd/`)
may import this module.  It is used by `tests/`, by `__graft_entry__.smoke()` and by the
`cpu_baseline` leg of `bench.py`, always as the checker / the timed CPU stand-in, never as
the thing that is shipped.

Parity status: PINNED.  `tests/golden/make_fixtures.py` imports the reference's own classes from
/root/reference (survey container only), drives them with the Philox random tape defined
below, and stores their outputs in `tests/golden/*.npz|json`; `tests/test_oracle_golden.py`
checks every function here against those vectors at float64 round-off.

Reference (paths under /root/reference):
  REG = multicore-pt-regression/pt_timeseries_regression.py
  CLS = multicore-pt-classification/pt_classification.py

Every function cites the REG/CLS lines it restates.  The code is written from the behaviour
tables in SURVEY.md section 8a (quirks Q1..Q14), vectorised over data rows where the reference
loops over them; `faithful=True` switches the forward pass to the reference's per-row cost
structure (one tiny dot product per row) so that the timed CPU baseline has the same shape of
work as the reference itself.
"""
from __future__ import annotations

import math
import numpy as np

TASK_REG = 0   # Gaussian likelihood, eta = log tau^2 sampled (REG)
TASK_CLS = 1   # multinomial likelihood (CLS)

# ---------------------------------------------------------------------------------------------
# Counter-based random tape (Philox4x32-10).  Shared specification with the HIP kernels:
#   key     = (seed & 0xffffffff, seed >> 32)
#   counter = (c0, c1, c2, c3) = (index, step-or-round, global replica id, stream id)
#   streams : 0 = per-step scalars (x0 -> lx, x1 -> MH u, (x2,x3) -> eta normal via cos branch)
#             1 = proposal noise for w  (c0 = j//4, element j%4 of the Box-Muller quad)
#             2 = swap uniforms         (c0 = pair k, c1 = swap round, c2 = 0)
#             3 = initial weights w0    (c0 = j//4, c1 = 0)
#   uniform : u = ((x >> 9) + 0.5) * 2^-23   (23 bits: exact in fp32 and fp64, never 0 or 1)
#   normals : quad (x0,x1,x2,x3) -> r0 = sqrt(-2 ln u(x0)), n0 = r0 cos(2 pi u(x1)), n1 = r0 sin(2 pi u(x1)),
#             r1 = sqrt(-2 ln u(x2)), n2 = r1 cos(2 pi u(x3)), n3 = r1 sin(2 pi u(x3))
# ---------------------------------------------------------------------------------------------
PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = 0x9E3779B9
PHILOX_W1 = 0xBB67AE85
STREAM_STEP, STREAM_WNOISE, STREAM_SWAP, STREAM_INIT = 0, 1, 2, 3
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, seed):
    """Philox4x32-10 on broadcastable uint arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3)))
    c0, c1, c2, c3 = c0.copy(), c1.copy(), c2.copy(), c3.copy()
    k0 = int(seed) & 0xFFFFFFFF
    k1 = (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0 = PHILOX_M0 * c0
        p1 = PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + PHILOX_W0) & 0xFFFFFFFF
        k1 = (k1 + PHILOX_W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def u24(x):
    """23-bit uniform in (0,1); the name is historical."""
    return ((np.asarray(x, dtype=np.uint32) >> np.uint32(9)).astype(np.float64) + 0.5) * (1.0 / 8388608.0)


def normal_quads(x0, x1, x2, x3):
    r0 = np.sqrt(-2.0 * np.log(u24(x0)))
    t0 = 2.0 * np.pi * u24(x1)
    r1 = np.sqrt(-2.0 * np.log(u24(x2)))
    t1 = 2.0 * np.pi * u24(x3)
    return np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], axis=-1)


class PhiloxTape:
    """The random tape every consumer (reference under patch, oracle, HIP kernel) reads."""

    def __init__(self, seed):
        self.seed = int(seed)

    def step_scalars(self, replica, step):
        """-> (lx, u_accept, n_eta) for MH step `step` of global replica `replica`."""
        x = philox4x32(0, step, replica, STREAM_STEP, self.seed)
        lx = float(u24(x[0]))
        u = float(u24(x[1]))
        n_eta = float(np.sqrt(-2.0 * np.log(u24(x[2]))) * np.cos(2.0 * np.pi * u24(x[3])))
        return lx, u, n_eta

    def _normals(self, n, c1, c2, stream):
        nq = (n + 3) // 4
        x = philox4x32(np.arange(nq), c1, c2, stream, self.seed)
        return normal_quads(*x).reshape(-1)[:n]

    def w_noise(self, replica, step, n):
        return self._normals(n, step, replica, STREAM_WNOISE)

    def w_init(self, replica, n):
        return self._normals(n, 0, replica, STREAM_INIT)

    def swap_uniforms(self, rnd, npairs):
        x = philox4x32(np.arange(npairs), rnd, 0, STREAM_SWAP, self.seed)
        return u24(x[0])


# ---------------------------------------------------------------------------------------------
# R1  decode / encode  (REG:80-97, CLS:85-106): w = [W1 (I x H row-major) | W2 (H x O) | B1 (H) | B2 (O)]
# ---------------------------------------------------------------------------------------------
def num_param(topo):
    I, H, O = topo
    return I * H + H * O + H + O


def decode(w, topo):
    I, H, O = topo
    a = I * H
    b = a + H * O
    return w[:a].reshape(I, H), w[a:b].reshape(H, O), w[b:b + H], w[b + H:b + H + O]


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# R2/R3  ForwardPass over all rows (REG:51-55,120-134; CLS:49-55,134-153).  Bias is SUBTRACTED (Q1),
# sigmoid on the output layer too (Q2).
def forward(X, w, topo, faithful=False):
    W1, W2, B1, B2 = decode(w, topo)
    if not faithful:
        hid = sigmoid(X @ W1 - B1)
        out = sigmoid(hid @ W2 - B2)
        return hid, out
    N = X.shape[0]
    hid = np.empty((N, topo[1]))
    out = np.empty((N, topo[2]))
    for n in range(N):                       # the reference's cost structure: one tiny dot per row
        h = sigmoid(X[n].dot(W1) - B1)
        hid[n] = h
        out[n] = sigmoid(h.dot(W2) - B2)
    return hid, out


# R4/R5  langevin_gradient: one sequential SGD epoch in file order (REG:57-78,99-118; CLS:72-82,114-132).
# delta_o = (t - out) out (1-out); delta_h = (delta_o . W2^T) hid (1-hid) with the PRE-update W2 (Q4);
# W2 += lr hid (x) delta_o; B2 -= lr delta_o; W1 += lr x (x) delta_h; B1 -= lr delta_h.
# CLS target = one-hot(int(y)); REG target = y (O == 1).
def langevin_gradient(data, w, topo, lr, task, depth=1):
    I, H, O = topo
    w = np.array(w, dtype=np.float64, copy=True)
    W1, W2, B1, B2 = decode(w, topo)          # views into w: in-place update like the reference
    X = data[:, :I]
    Y = data[:, I]
    for _ in range(depth):
        for n in range(data.shape[0]):
            x = X[n]
            hid = sigmoid(x.dot(W1) - B1)
            out = sigmoid(hid.dot(W2) - B2)
            if task == TASK_CLS:
                t = np.zeros(O)
                t[int(Y[n])] = 1.0
            else:
                t = data[n, I:I + O]
            od = (t - out) * (out * (1.0 - out))
            hd = od.dot(W2.T) * (hid * (1.0 - hid))
            W2 += lr * np.outer(hid, od)
            B2 -= lr * od
            W1 += lr * np.outer(x, hd)
            B1 -= lr * hd
    return w


def rmse(pred, actual):
    return float(np.sqrt(((pred - actual) ** 2).mean()))


# R6  likelihood_func.  REG:200-205 Gaussian; CLS:209-222 multinomial on softmax-of-sigmoid (Q3).
# Returns (tempered loglik, fx, rmse[, accuracy]).
def likelihood_reg(data, w, tau_sq, topo, adapttemp, faithful=False):
    I = topo[0]
    y = data[:, I]
    fx = forward(data[:, :I], w, topo, faithful)[1][:, 0]
    loss = np.sum(-0.5 * np.log(2.0 * math.pi * tau_sq) - 0.5 * np.square(y - fx) / tau_sq)
    return float(loss) / adapttemp, fx, rmse(fx, y)


def likelihood_cls(data, w, topo, adapttemp, faithful=False):
    I = topo[0]
    y = data[:, I]
    out = forward(data[:, :I], w, topo, faithful)[1]
    fx = np.argmax(out, axis=1).astype(np.float64)
    e = np.exp(out)
    prob = e / e.sum(axis=1, keepdims=True)
    lhood = float(np.sum(np.log(prob[np.arange(data.shape[0]), y.astype(np.int64)])))
    return lhood / adapttemp, fx, rmse(fx, y)


def accuracy(pred, actual):                    # CLS:200-207
    return 100.0 * (float(np.count_nonzero(pred == actual)) / pred.shape[0])


# R7  prior_likelihood.  REG:215-221 (constant uses d*h+h+2, Q5); CLS:224-230.
def prior_reg(sigma_squared, nu_1, nu_2, w, tausq, topo):
    d, h = topo[0], topo[1]
    part1 = -1 * ((d * h + h + 2) / 2) * np.log(sigma_squared)
    part2 = 1 / (2 * sigma_squared) * float(np.sum(np.square(w)))
    return part1 - part2 - (1 + nu_1) * np.log(tausq) - (nu_2 / tausq)


def prior_cls(sigma_squared, w, topo):
    d, h, o = topo
    part1 = -1 * ((d * h + h + o + h * o) / 2) * np.log(sigma_squared)
    part2 = 1 / (2 * sigma_squared) * float(np.sum(np.square(w)))
    return part1 - part2


# R13  ladder (REG:529-636 with the only call used: ndim=2, ntemps and Tmax given).
def temperature_ladder(num_chains, maxtemp):
    betas = np.logspace(0, -np.log10(maxtemp), num_chains)
    return [float(1.0 / b) for b in betas]


# R12  swap cascade (REG:659-690,741-748): sequential bubble pass, closed form of SURVEY 3.3.
def swap_cascade(L, u):
    """L[R] posted scalars, u[R-1] uniforms -> (src[R], n_swapped). slot k receives state of slot src[k]."""
    R = len(L)
    src = [0] * R
    c = 0
    nsw = 0
    for k in range(R - 1):
        d = L[k + 1] - L[c]
        d = 709 if not (d < 709) else d        # python min(709, nan) == 709
        try:
            p = min(1, 0.5 * math.exp(d))
        except OverflowError:
            p = 1
        if u[k] < p:
            src[k] = k + 1
            nsw += 1
        else:
            src[k] = c
            c = k + 1
    src[R - 1] = c
    return src, nsw


def swap_trigger(task, i, swap_interval):
    """Q10: REG hands off after step i when i % si == 0 and i != 0 (REG:427); CLS when (i+1) % si == 0 (CLS:438)."""
    if task == TASK_REG:
        return i % swap_interval == 0 and i != 0
    return (i + 1) % swap_interval == 0


def count_handoffs(task, S, swap_interval):
    return sum(1 for i in range(S - 1) if swap_trigger(task, i, swap_interval))


class Replica:
    """One chain: the state and loop body of ptReplica.run (REG:223-447, CLS:232-456)."""

    def __init__(self, task, topo, train, test, w0, temperature, samples, use_lg, l_prob, lr,
                 tape, gid, faithful=False, step_w=0.025, step_eta=0.2, sigma_squared=25.0, nu_1=0.0, nu_2=0.0):
        self.task, self.topo = task, tuple(topo)
        self.train, self.test = train, test
        self.T = float(temperature)
        self.adapttemp = self.T
        self.S = int(samples)
        self.use_lg, self.l_prob, self.lr = bool(use_lg), float(l_prob), float(lr)
        self.tape, self.gid, self.faithful = tape, int(gid), faithful
        self.noise_gid = int(gid)                          # Q14 option: PTOracle(shared_noise=True) points every chain at tape 0
        self.step_w, self.step_eta = step_w, step_eta
        self.sigma_squared, self.nu_1, self.nu_2 = sigma_squared, nu_1, nu_2
        P = num_param(topo)
        self.P = P
        S = self.S
        self.pos_w = np.ones((S, P))                       # Q7: row 0 = ones
        self.rmse_train = np.zeros(S)
        self.rmse_test = np.zeros(S)
        self.acc_train = np.zeros(S)
        self.acc_test = np.zeros(S)
        self.likeh = np.zeros((S, 2))
        self.likeh[0, :] = [-100, -100]
        self.accept_list = np.zeros(S)
        self.num_accepted = 0
        self.langevin_count = 0
        self.w = np.array(w0, dtype=np.float64)
        self.pt_samples = S * 0.6
        self.init_count = 0
        self.lik_stale = False                             # Q12: (w, eta) arrived through a swap, likelihood / prior still the old state's
        I = topo[0]
        self.y_train, self.y_test = train[:, I], test[:, I]
        if task == TASK_REG:                               # R14 (REG:266-285)
            pred_train = forward(train[:, :I], self.w, topo)[1][:, 0]
            self.eta = float(np.log(np.var(pred_train - self.y_train)))
            self.tau_pro = float(np.exp(self.eta))
            self.prior_current = prior_reg(sigma_squared, nu_1, nu_2, self.w, self.tau_pro, topo)
            self.likelihood = likelihood_reg(train, self.w, self.tau_pro, topo, self.adapttemp)[0]
        else:                                              # CLS:271-284
            self.eta = 0.0
            self.tau_pro = 1.0
            self.prior_current = prior_cls(sigma_squared, self.w, topo)
            self.likelihood = likelihood_cls(train, self.w, topo, self.adapttemp)[0]

    def _lik(self, data, w, tau):
        if self.task == TASK_REG:
            return likelihood_reg(data, w, tau, self.topo, self.adapttemp, self.faithful)
        return likelihood_cls(data, w, self.topo, self.adapttemp, self.faithful)

    def step(self, i, force=None):
        """Loop body for index i (REG:313-423 / CLS:313-434).  force = None: the chain decides for itself (the reference's
        behaviour); True / False: the decision is imposed -- used by tests that FOLLOW the device's chain past a decision the
        two sides took differently inside the fp32 error of log alpha; `last_natural` keeps what this chain would have done."""
        if i < self.pt_samples:
            self.adapttemp = self.T
        if i == self.pt_samples and self.init_count == 0:   # R10/Q9: stale tau_pro, float-equality trigger
            self.adapttemp = 1
            self.likelihood = self._lik(self.train, self.w, self.tau_pro)[0]
            self.init_count = 1
            self.lik_stale = False                          # re-evaluated on the current w (the prior is not: REG:322-324)
        lx, u, n_eta = self.tape.step_scalars(self.noise_gid, i)
        noise = self.tape.w_noise(self.noise_gid, i, self.P)
        if self.use_lg and lx < self.l_prob:
            w_gd = langevin_gradient(self.train, self.w, self.topo, self.lr, self.task)
            w_proposal = w_gd + self.step_w * noise
            w_prop_gd = langevin_gradient(self.train, w_proposal, self.topo, self.lr, self.task)
            wc_delta = self.w - w_prop_gd
            wp_delta = w_proposal - w_gd
            sigma_sq = self.step_w * self.step_w
            first = -0.5 * np.sum(wc_delta * wc_delta) / sigma_sq
            second = -0.5 * np.sum(wp_delta * wp_delta) / sigma_sq
            diff_prop = (first - second) / self.adapttemp    # Q6
            self.langevin_count += 1
        else:
            diff_prop = 0
            w_proposal = self.w + self.step_w * noise
        if self.task == TASK_REG:
            eta_pro = self.eta + self.step_eta * n_eta
            self.tau_pro = math.exp(eta_pro)
        else:
            eta_pro = self.eta
        lik_prop, pred_train, rmsetrain = self._lik(self.train, w_proposal, self.tau_pro)
        _, pred_test, rmsetest = self._lik(self.test, w_proposal, self.tau_pro)
        if self.task == TASK_REG:
            prior_prop = prior_reg(self.sigma_squared, self.nu_1, self.nu_2, w_proposal, self.tau_pro, self.topo)
        else:
            prior_prop = prior_cls(self.sigma_squared, w_proposal, self.topo)
        diff_prior = prior_prop - self.prior_current
        diff_likelihood = lik_prop - self.likelihood
        try:
            mh_prob = min(1, math.exp(diff_likelihood + diff_prior + diff_prop))   # Q8: nan -> 1
        except OverflowError:
            mh_prob = 1
        self.accept_list[i + 1] = self.num_accepted
        self.likeh[i + 1, 0] = lik_prop if self.task == TASK_REG else lik_prop * self.adapttemp
        self.last_logalpha = diff_likelihood + diff_prior + diff_prop
        self.last_u = u
        self.last_stale = self.lik_stale
        # size of the terms log alpha is the small difference of (the parity tests state fp32 error bounds relative to it)
        self.last_scale = (abs(lik_prop) + abs(self.likelihood) + abs(prior_prop) + abs(self.prior_current) +
                           ((abs(first) + abs(second)) / self.adapttemp if diff_prop != 0 else 0.0))
        self.last_natural = bool(u < mh_prob)
        take = self.last_natural if force is None else bool(force)
        self.last_forced = take != self.last_natural
        if take:
            self.num_accepted += 1
            self.lik_stale = False
            self.likelihood = lik_prop
            self.prior_current = prior_prop
            self.w = w_proposal
            self.eta = eta_pro
            if self.task == TASK_CLS:
                self.acc_train[i + 1] = accuracy(pred_train, self.y_train)
                self.acc_test[i + 1] = accuracy(pred_test, self.y_test)
            self.pos_w[i + 1] = w_proposal
            self.rmse_train[i + 1] = rmsetrain
            self.rmse_test[i + 1] = rmsetest
            return True
        self.pos_w[i + 1] = self.pos_w[i]
        self.rmse_train[i + 1] = self.rmse_train[i]
        self.rmse_test[i + 1] = self.rmse_test[i]
        self.acc_train[i + 1] = self.acc_train[i]
        self.acc_test[i + 1] = self.acc_test[i]
        return False

    def posted_L(self):
        """Q11: REG posts likelihood*T (REG:430), CLS posts the tempered likelihood (CLS:439)."""
        return self.likelihood * self.T if self.task == TASK_REG else self.likelihood


class PTOracle:
    """ParallelTempering.run_chains restated for one process (REG:694-771, CLS:701-776).

    `first`/`count` select a contiguous shard of the ladder (used by the sharded tests); swap
    rounds then need `exchange` to be driven from outside (see tests/test_sharding_gloo.py).
    """

    def __init__(self, task, topo, train, test, num_chains, maxtemp, NumSample, swap_interval,
                 use_lg=False, l_prob=0.5, lr=0.1, seed=1, w0=None, faithful=False, swap_rule=0, shared_noise=False,
                 label_swap=False):
        self.swap_rule = swap_rule          # 0 = the reference's cascade; 1 = even/odd Metropolis (NOT in the reference)
        # label swapping (NOT in the reference; parity unpinned): a round permutes which CHAIN holds which
        # temperature instead of moving (w, eta) between temperature slots.  self.replicas stays ordered by temperature index;
        # a chain keeps everything it owns (likelihood, prior, counters, its noise stream) and its likelihood is re-tempered.
        self.label_swap = bool(label_swap)
        self.holder_hist = []               # [(first trace row written under this assignment, [chain objects by temperature])]
        self.task, self.topo = task, tuple(topo)
        self.train = np.asarray(train, dtype=np.float64)
        self.test = np.asarray(test, dtype=np.float64)
        self.R = int(num_chains)
        self.S = int(NumSample / num_chains)
        self.si = int(swap_interval)
        self.tape = PhiloxTape(seed)
        self.temperatures = temperature_ladder(num_chains, maxtemp)
        P = num_param(topo)
        self.P = P
        if w0 is None:
            w0 = np.stack([self.tape.w_init(r, P) for r in range(self.R)])
        self.replicas = [Replica(task, topo, self.train, self.test, w0[r], self.temperatures[r], self.S,
                                 use_lg, l_prob, lr, self.tape, r, faithful) for r in range(self.R)]
        if shared_noise:                                   # Q14: forked chains inherit ONE RNG state (REG:709-712)
            for rep in self.replicas:
                rep.noise_gid = 0
        self.num_swap = 0
        self.total_swap_proposals = 0
        self.rounds_done = 0
        self.src_log = []
        self.holder_hist.append((0, list(self.replicas)))
        self._steps_done = 0
        self.L_log = []                                     # the scalars every round decided on (swap_rule 0)

    def swap_round_even_odd(self):
        """Option swap_rule = 1, parity unpinned: the reference has no such rule; this restates the textbook
        exchange of adjacent pairs with probability min(1, exp((1/T_k - 1/T_k+1)(L_k+1 - L_k))) on untempered
        log-likelihoods, pairs of alternating parity per round; a moved state brings its likelihood (re-tempered for its
        new slot) and prior along."""
        R = self.R
        u = self.tape.swap_uniforms(self.rounds_done, R - 1)
        raw = [rep.likelihood * rep.adapttemp for rep in self.replicas]
        T = self.temperatures
        src = list(range(R))
        par = self.rounds_done & 1
        nsw = 0
        for k in range(par, R - 1, 2):
            d = (1.0 / T[k] - 1.0 / T[k + 1]) * (raw[k + 1] - raw[k])
            p = 1.0 if d != d else min(1.0, math.exp(min(d, 80.0)))
            if u[k] < p:
                src[k], src[k + 1] = k + 1, k
                nsw += 1
        self.num_swap += nsw
        self.total_swap_proposals += (R - par) // 2
        self.rounds_done += 1
        self.src_log.append(list(src))
        if self.label_swap:
            self._apply_labels(src)
            return src
        old = [(rep.w, rep.eta, raw[i], rep.prior_current) for i, rep in enumerate(self.replicas)]
        for k, rep in enumerate(self.replicas):
            if src[k] != k:
                w, eta, lraw, pri = old[src[k]]
                rep.w, rep.eta, rep.prior_current = w, eta, pri
                rep.likelihood = lraw / rep.adapttemp
        return src

    def _apply_labels(self, src):
        """Temperature t is handed to the chain that held temperature src[t]."""
        T = self.temperatures
        new = [self.replicas[src[t]] for t in range(self.R)]
        for t, rep in enumerate(new):
            if rep.T != T[t]:
                if rep.init_count == 0:                     # still tempered: the chain's likelihood follows its new temperature
                    rep.likelihood *= rep.T / T[t]
                rep.T = float(T[t])
        self.replicas = new
        self.holder_hist.append((self._steps_done + 1, list(new)))

    def traces_by_temperature(self):
        """Label swapping: what the per-temperature files hold -- for every temperature the rows recorded by the chain that
        held it at the time.  -> dict of arrays [R, S, ...]."""
        out = dict(pos_w=np.zeros((self.R, self.S, self.P)), likeh=np.zeros((self.R, self.S)), accept=np.zeros((self.R, self.S)),
                   rmse_train=np.zeros((self.R, self.S)), rmse_test=np.zeros((self.R, self.S)),
                   acc_train=np.zeros((self.R, self.S)), acc_test=np.zeros((self.R, self.S)))
        bounds = [h[0] for h in self.holder_hist] + [self.S]
        for (row0, chains), row1 in zip(self.holder_hist, bounds[1:]):
            for t, rep in enumerate(chains):
                out["pos_w"][t, row0:row1] = rep.pos_w[row0:row1]
                out["likeh"][t, row0:row1] = rep.likeh[row0:row1, 0]
                out["accept"][t, row0:row1] = rep.accept_list[row0:row1]
                for k in ("rmse_train", "rmse_test", "acc_train", "acc_test"):
                    out[k][t, row0:row1] = getattr(rep, k)[row0:row1]
        return out

    def swap_round(self, L=None, apply=True, force_src=None):
        """force_src: impose this permutation instead of the cascade's own (tests that follow the device's chain; the caller has
        checked every pair decision in which the two differ); the scalars the round decided on are logged either way."""
        if self.swap_rule == 1:
            return self.swap_round_even_odd()
        R = self.R
        if L is None:
            L = [rep.posted_L() for rep in self.replicas]
        u = self.tape.swap_uniforms(self.rounds_done, R - 1)
        src, nsw = swap_cascade(L, u)
        self.last_natural_src = list(src)
        if force_src is not None:
            src = [int(v) for v in force_src]
            nsw = sum(1 for k in range(R - 1) if src[k] == k + 1)
        self.num_swap += nsw
        self.total_swap_proposals += R - 1
        self.rounds_done += 1
        self.src_log.append(list(src))
        self.L_log.append([float(v) for v in L])
        if apply and self.label_swap:
            self._apply_labels(src)
        elif apply:                                         # R11/Q12: only (w, eta) move; likelihood/prior stay stale
            ws = [rep.w for rep in self.replicas]
            etas = [rep.eta for rep in self.replicas]
            for k, rep in enumerate(self.replicas):
                rep.w = ws[src[k]]
                rep.eta = etas[src[k]]
                if src[k] != k:
                    rep.lik_stale = True
        return src

    def run(self):
        S, si = self.S, self.si
        for i in range(S - 1):
            for rep in self.replicas:
                rep.step(i)
            self._steps_done = i + 1
            if swap_trigger(self.task, i, si):
                self.swap_round()
        # Q13 phantom round: the parent loops int(S/si) rounds; extra ones consume the end-of-chain
        # vectors (L = final tempered likelihood, REG:442 / CLS:451) and are counted but discarded.
        rounds = int(S / si) if si > 0 else 0
        if self.swap_rule == 0 and rounds > self.rounds_done:
            self.swap_round(L=[rep.likelihood for rep in self.replicas], apply=False)
        return self

    @property
    def swap_perc(self):
        return self.num_swap * 100 / self.total_swap_proposals
