/*
 * ptnn_oracle_c.c -- CPU oracle in plain C: float64 restatement of the reference's parallel-tempering hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (parallel-tempering-neural-net_amd/) may link, load or call this
 * file.  It exists because the numpy oracle (oracle/ptnn_oracle.py, the readable restatement) needs milliseconds per MH step:
 * too slow to FOLLOW the device through a whole run of the reference's standard length (64 chains x 10 000 samples).  This file
 * restates the same functions at a few tens of microseconds per step; tests/test_oracle_golden.py holds it to the same
 * reference-generated golden vectors (tests/golden/, F1-F4) as the numpy oracle and to the numpy oracle itself.
 *
 * Parity status: PINNED through those fixtures (made by tests/golden/make_fixtures.py importing the reference itself).
 *
 * Reference (paths under /root/reference):
 *   REG = multicore-pt-regression/pt_timeseries_regression.py
 *   CLS = multicore-pt-classification/pt_classification.py
 * Every function cites the REG/CLS lines it restates (written from SURVEY.md section 8a, quirks Q1..Q14).
 *
 * Build: gcc -O2 -fPIC -shared -o oracle/libptnn_oracle.so oracle/ptnn_oracle_c.c -lm   (oracle/Makefile, __graft_entry__.build)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TASK_REG 0
#define TASK_CLS 1

/* ---- random tape: Philox4x32-10, same specification as oracle/ptnn_oracle.py and the HIP kernels ---- */
static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint64_t seed, uint32_t x[4]) {
    uint32_t k0 = (uint32_t)(seed & 0xffffffffu), k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
}
static double u23(uint32_t x) { return ((double)(x >> 9) + 0.5) * (1.0 / 8388608.0); }

void orc_step_scalars(uint64_t seed, int replica, int step, double out[3]) {       /* -> lx, u, n_eta */
    uint32_t x[4];
    philox4x32_10(0u, (uint32_t)step, (uint32_t)replica, 0u, seed, x);
    out[0] = u23(x[0]);
    out[1] = u23(x[1]);
    out[2] = sqrt(-2.0 * log(u23(x[2]))) * cos(2.0 * M_PI * u23(x[3]));
}
static void normals(uint64_t seed, uint32_t c1, uint32_t c2, uint32_t stream, int n, double* out) {
    for (int q = 0; 4 * q < n; ++q) {
        uint32_t x[4];
        philox4x32_10((uint32_t)q, c1, c2, stream, seed, x);
        const double r0 = sqrt(-2.0 * log(u23(x[0]))), t0 = 2.0 * M_PI * u23(x[1]);
        const double r1 = sqrt(-2.0 * log(u23(x[2]))), t1 = 2.0 * M_PI * u23(x[3]);
        const double v[4] = {r0 * cos(t0), r0 * sin(t0), r1 * cos(t1), r1 * sin(t1)};
        for (int k = 0; k < 4 && 4 * q + k < n; ++k) out[4 * q + k] = v[k];
    }
}
void orc_w_noise(uint64_t seed, int replica, int step, int n, double* out) { normals(seed, (uint32_t)step, (uint32_t)replica, 1u, n, out); }
void orc_w_init(uint64_t seed, int replica, int n, double* out) { normals(seed, 0u, (uint32_t)replica, 3u, n, out); }

/* ---- the network: w = [W1 (I x H row-major) | W2 (H x O) | B1 (H) | B2 (O)]  (R1: REG:80-97, CLS:85-106) ---- */
typedef struct { int task, I, H, O; } orc_topo;

static double sigm(double x) { return 1.0 / (1.0 + exp(-x)); }

/* R2 ForwardPass on one row (REG:51-55, CLS:49-55): bias SUBTRACTED (Q1), sigmoid on the output layer too (Q2) */
static void forward_row(const orc_topo* t, const double* w, const double* x, double* hid, double* out) {
    const int I = t->I, H = t->H, O = t->O;
    const double *W1 = w, *W2 = w + I * H, *B1 = W2 + H * O, *B2 = B1 + H;
    for (int h = 0; h < H; ++h) {
        double z = 0.0;
        for (int i = 0; i < I; ++i) z += x[i] * W1[i * H + h];
        hid[h] = sigm(z - B1[h]);
    }
    for (int o = 0; o < O; ++o) {
        double z = 0.0;
        for (int h = 0; h < H; ++h) z += hid[h] * W2[h * O + o];
        out[o] = sigm(z - B2[o]);
    }
}

/* R4/R5 langevin_gradient: one sequential SGD epoch in file order (REG:57-78, 99-118; CLS:72-82, 114-132).
 * delta_o = (t - out) out (1 - out); delta_h = (delta_o . W2^T) hid (1 - hid) with the PRE-update W2 (Q4);
 * W2 += lr hid (x) delta_o; B2 -= lr delta_o; W1 += lr x (x) delta_h; B1 -= lr delta_h.  CLS target = one-hot(int(y)). */
void orc_langevin_gradient(int task, int I, int H, int O, const double* data, int n_rows, int ncols, const double* w_in, double lr,
                           double* w_out, double* scratch /* 2 H + 2 O */) {
    const orc_topo t = {task, I, H, O};
    const int P = I * H + H * O + H + O;
    if (w_out != w_in) memcpy(w_out, w_in, (size_t)P * sizeof(double));
    double *W1 = w_out, *W2 = w_out + I * H, *B1 = W2 + H * O, *B2 = B1 + H;
    double *hid = scratch, *out = hid + H, *od = out + O, *hd = od + O;
    for (int n = 0; n < n_rows; ++n) {
        const double* x = data + (size_t)n * ncols;
        forward_row(&t, w_out, x, hid, out);
        for (int o = 0; o < O; ++o) {
            const double tgt = (task == TASK_CLS) ? ((int)x[I] == o ? 1.0 : 0.0) : x[I + o];
            od[o] = (tgt - out[o]) * (out[o] * (1.0 - out[o]));
        }
        for (int h = 0; h < H; ++h) {
            double g = 0.0;
            for (int o = 0; o < O; ++o) g += od[o] * W2[h * O + o];
            hd[h] = g * (hid[h] * (1.0 - hid[h]));
        }
        for (int h = 0; h < H; ++h)
            for (int o = 0; o < O; ++o) W2[h * O + o] += lr * hid[h] * od[o];
        for (int o = 0; o < O; ++o) B2[o] -= lr * od[o];
        for (int i = 0; i < I; ++i)
            for (int h = 0; h < H; ++h) W1[i * H + h] += lr * x[i] * hd[h];
        for (int h = 0; h < H; ++h) B1[h] -= lr * hd[h];
    }
}

/* R3 + R6 evaluate_proposal + likelihood_func (REG:120-134, 200-205; CLS:134-153, 209-222).  Returns the TEMPERED log-likelihood;
 * *rmse_out: REG sqrt(mean((fx - y)^2)), CLS the same between predicted class id and label; *acc_out: CLS 100 * matches / N
 * (CLS:200-207).  CLS: prob = softmax of the already sigmoided outputs (Q3), argmax = first maximum. fx_out may be NULL. */
double orc_likelihood(int task, int I, int H, int O, const double* data, int n_rows, int ncols, const double* w, double tau_sq,
                      double adapttemp, double* rmse_out, double* acc_out, double* fx_out, double* scratch /* H + O */) {
    const orc_topo t = {task, I, H, O};
    double *hid = scratch, *out = hid + H;
    double loss = 0.0, se = 0.0;
    int match = 0;
    for (int n = 0; n < n_rows; ++n) {
        const double* x = data + (size_t)n * ncols;
        const double y = x[I];
        forward_row(&t, w, x, hid, out);
        if (task == TASK_REG) {
            const double fx = out[0];
            loss += -0.5 * log(2.0 * M_PI * tau_sq) - 0.5 * ((y - fx) * (y - fx)) / tau_sq;
            se += (fx - y) * (fx - y);
            if (fx_out) fx_out[n] = fx;
        } else {
            int best = 0;
            double den = 0.0;
            for (int o = 0; o < O; ++o) {
                if (out[o] > out[best]) best = o;
                den += exp(out[o]);
            }
            loss += log(exp(out[(int)y]) / den);
            se += ((double)best - y) * ((double)best - y);
            if ((double)best == y) ++match;
            if (fx_out) fx_out[n] = (double)best;
        }
    }
    if (rmse_out) *rmse_out = sqrt(se / n_rows);
    if (acc_out) *acc_out = 100.0 * ((double)match / n_rows);
    return loss / adapttemp;
}

/* R7 prior_likelihood.  REG:215-221: constant term uses d*h + h + 2 (Q5); CLS:224-230.  Not tempered. */
double orc_prior(int task, int I, int H, int O, double sigma_sq, double nu1, double nu2, const double* w, double tau_sq) {
    const int P = I * H + H * O + H + O;
    double ss = 0.0;
    for (int j = 0; j < P; ++j) ss += w[j] * w[j];
    const double part2 = 1.0 / (2.0 * sigma_sq) * ss;
    if (task == TASK_REG) {
        const double part1 = -1.0 * ((I * H + H + 2) / 2.0) * log(sigma_sq);
        return part1 - part2 - (1.0 + nu1) * log(tau_sq) - (nu2 / tau_sq);
    }
    const double part1 = -1.0 * ((I * H + H + O + H * O) / 2.0) * log(sigma_sq);
    return part1 - part2;
}

/* ---- one chain: the state and loop body of ptReplica.run (REG:223-447, CLS:232-456).  Every array is owned by the caller
 * (numpy, see oracle/ptnn_oracle_c.py); the layout is mirrored by a ctypes.Structure there. ---- */
typedef struct {
    int32_t task, I, H, O, P, Ntr, Nte, ncols, S, use_lg, gid, noise_gid;
    int32_t num_accepted, langevin_count, init_count, lik_stale;
    int32_t last_stale, last_natural, last_forced, last_lg;        /* last_lg: the step drew a Langevin proposal */
    uint64_t seed;
    double T, adapttemp, l_prob, lr, step_w, step_eta, sigma_sq, nu1, nu2, pt_samples;
    double eta, tau_pro, likelihood, prior_current;
    double last_logalpha, last_u, last_scale;
    double last_lik_cur, last_prior_cur;        /* the cached likelihood / prior the step's log alpha was formed with (after the R10 switch) */
    const double *train, *test;                 /* [N][ncols] row-major */
    double* w;                                  /* [P] current state */
    double *pos_w, *likeh, *accept_list, *rmse_train, *rmse_test, *acc_train, *acc_test;   /* traces: [S][P], [S][2], [S] ... */
    double* scratch;                            /* [4 P + 2 H + 2 O] */
} orc_replica;

/* R14 chain start-up (REG:266-285, CLS:271-284) */
void orc_replica_init(orc_replica* r) {
    double* sc = r->scratch + 4 * r->P;
    r->adapttemp = r->T;
    if (r->task == TASK_REG) {
        /* eta = log var(pred_train - y_train) (REG:270, np.var: population variance) */
        double mean = 0.0, m2 = 0.0;
        double* fx = (double*)malloc((size_t)r->Ntr * sizeof(double));
        (void)orc_likelihood(r->task, r->I, r->H, r->O, r->train, r->Ntr, r->ncols, r->w, 1.0, 1.0, NULL, NULL, fx, sc);
        for (int n = 0; n < r->Ntr; ++n) mean += fx[n] - r->train[(size_t)n * r->ncols + r->I];
        mean /= r->Ntr;
        for (int n = 0; n < r->Ntr; ++n) { const double d = fx[n] - r->train[(size_t)n * r->ncols + r->I] - mean; m2 += d * d; }
        free(fx);
        r->eta = log(m2 / r->Ntr);
        r->tau_pro = exp(r->eta);
    } else {
        r->eta = 0.0;
        r->tau_pro = 1.0;
    }
    r->prior_current = orc_prior(r->task, r->I, r->H, r->O, r->sigma_sq, r->nu1, r->nu2, r->w, r->tau_pro);
    r->likelihood = orc_likelihood(r->task, r->I, r->H, r->O, r->train, r->Ntr, r->ncols, r->w, r->tau_pro, r->adapttemp, NULL, NULL, NULL, sc);
}

/* Loop body for index i (REG:313-423 / CLS:313-434).  force < 0: the chain decides for itself; force = 0 / 1: the decision is
 * imposed (the caller follows another implementation's chain and has checked that the two decisions differ only inside the
 * fp32 error of log alpha); last_natural keeps what this chain would have decided.  Returns the decision taken. */
/* The chain's state replaced by another implementation's (w, eta) -- the fp32 values a device recorded for this chain -- with the
 * cached likelihood and prior RE-EVALUATED here in float64 from that state (what REG:395-399 keeps after an accepted step, at
 * the temperature and noise variance of that step).  A followed run (tests/parity.py: follow_device_run) calls this after every
 * accepted step, so that the next step's log alpha is compared between two sides that start from the SAME state: what is then
 * measured is the fp32 error of ONE step, not the drift of two chains over thousands of accepted steps.  w_new: [P];
 * eta_new: NaN = keep the chain's own. */
void orc_replica_set_state(orc_replica* r, const double* w_new, double eta_new) {
    double* sc = r->scratch + 4 * r->P;
    memcpy(r->w, w_new, (size_t)r->P * sizeof(double));
    if (eta_new == eta_new && r->task == TASK_REG) r->eta = eta_new;
    const double tau = (r->task == TASK_REG) ? exp(r->eta) : 1.0;
    r->likelihood = orc_likelihood(r->task, r->I, r->H, r->O, r->train, r->Ntr, r->ncols, r->w, tau, r->adapttemp, NULL, NULL, NULL, sc);
    r->prior_current = orc_prior(r->task, r->I, r->H, r->O, r->sigma_sq, r->nu1, r->nu2, r->w, tau);
    r->lik_stale = 0;
}

int orc_replica_step(orc_replica* r, int i, int force) {
    const int P = r->P, S = r->S;
    double *noise = r->scratch, *w_prop = noise + P, *w_gd = w_prop + P, *w_pgd = w_gd + P, *sc = w_pgd + P;
    if ((double)i < r->pt_samples) r->adapttemp = r->T;
    if ((double)i == r->pt_samples && r->init_count == 0) {        /* R10 / Q9: stale tau_pro, float-equality trigger (REG:320-324) */
        r->adapttemp = 1.0;
        r->likelihood = orc_likelihood(r->task, r->I, r->H, r->O, r->train, r->Ntr, r->ncols, r->w, r->tau_pro, r->adapttemp, NULL, NULL, NULL, sc);
        r->init_count = 1;
        r->lik_stale = 0;
    }
    double scal[3];
    orc_step_scalars(r->seed, r->noise_gid, i, scal);
    const double lx = scal[0], u = scal[1], n_eta = scal[2];
    orc_w_noise(r->seed, r->noise_gid, i, P, noise);
    double diff_prop = 0.0, first = 0.0, second = 0.0;
    if (r->use_lg && lx < r->l_prob) {                             /* REG:329-347 */
        orc_langevin_gradient(r->task, r->I, r->H, r->O, r->train, r->Ntr, r->ncols, r->w, r->lr, w_gd, sc);
        for (int j = 0; j < P; ++j) w_prop[j] = w_gd[j] + r->step_w * noise[j];
        orc_langevin_gradient(r->task, r->I, r->H, r->O, r->train, r->Ntr, r->ncols, w_prop, r->lr, w_pgd, sc);
        const double sig = r->step_w * r->step_w;
        double a = 0.0, b = 0.0;
        for (int j = 0; j < P; ++j) {
            const double dc = r->w[j] - w_pgd[j], dp = w_prop[j] - w_gd[j];
            a += dc * dc; b += dp * dp;
        }
        first = -0.5 * a / sig; second = -0.5 * b / sig;
        diff_prop = (first - second) / r->adapttemp;               /* Q6 */
        r->langevin_count += 1;
        r->last_lg = 1;
    } else {
        r->last_lg = 0;
        for (int j = 0; j < P; ++j) w_prop[j] = r->w[j] + r->step_w * noise[j];
    }
    double eta_pro = r->eta;
    if (r->task == TASK_REG) {                                     /* REG:355-356 */
        eta_pro = r->eta + r->step_eta * n_eta;
        r->tau_pro = exp(eta_pro);
    }
    double rm_tr, rm_te, ac_tr, ac_te;
    const double lik_prop = orc_likelihood(r->task, r->I, r->H, r->O, r->train, r->Ntr, r->ncols, w_prop, r->tau_pro, r->adapttemp, &rm_tr, &ac_tr, NULL, sc);
    (void)orc_likelihood(r->task, r->I, r->H, r->O, r->test, r->Nte, r->ncols, w_prop, r->tau_pro, r->adapttemp, &rm_te, &ac_te, NULL, sc);
    const double prior_prop = orc_prior(r->task, r->I, r->H, r->O, r->sigma_sq, r->nu1, r->nu2, w_prop, r->tau_pro);
    const double la = (lik_prop - r->likelihood) + (prior_prop - r->prior_current) + diff_prop;
    /* mh_prob = min(1, exp(la)); OverflowError -> 1; nan -> min(1, nan) == 1: NaN proposals are accepted (Q8, REG:372-376) */
    const double mh = (la != la) ? 1.0 : (la > 709.0 ? 1.0 : fmin(1.0, exp(la)));
    r->accept_list[i + 1] = (double)r->num_accepted;               /* Q7: the count BEFORE this step (REG:380) */
    r->likeh[2 * (i + 1)] = (r->task == TASK_REG) ? lik_prop : lik_prop * r->adapttemp;     /* REG:391 / CLS:404 */
    r->last_logalpha = la; r->last_u = u; r->last_stale = r->lik_stale;
    r->last_lik_cur = r->likelihood; r->last_prior_cur = r->prior_current;
    r->last_scale = fabs(lik_prop) + fabs(r->likelihood) + fabs(prior_prop) + fabs(r->prior_current) +
                    (diff_prop != 0.0 ? (fabs(first) + fabs(second)) / r->adapttemp : 0.0);
    const int natural = (u < mh) ? 1 : 0;
    const int take = (force < 0) ? natural : (force ? 1 : 0);
    r->last_natural = natural; r->last_forced = (take != natural);
    if (take) {                                                    /* REG:395-413 */
        r->num_accepted += 1;
        r->lik_stale = 0;
        r->likelihood = lik_prop;
        r->prior_current = prior_prop;
        memcpy(r->w, w_prop, (size_t)P * sizeof(double));
        r->eta = eta_pro;
        if (r->task == TASK_CLS) { r->acc_train[i + 1] = ac_tr; r->acc_test[i + 1] = ac_te; }
        memcpy(r->pos_w + (size_t)(i + 1) * P, w_prop, (size_t)P * sizeof(double));
        r->rmse_train[i + 1] = rm_tr;
        r->rmse_test[i + 1] = rm_te;
    } else {                                                       /* REG:416-423 */
        memcpy(r->pos_w + (size_t)(i + 1) * P, r->pos_w + (size_t)i * P, (size_t)P * sizeof(double));
        r->rmse_train[i + 1] = r->rmse_train[i];
        r->rmse_test[i + 1] = r->rmse_test[i];
        r->acc_train[i + 1] = r->acc_train[i];
        r->acc_test[i + 1] = r->acc_test[i];
    }
    (void)S;
    return take;
}

/* steps [i0, i1) of one chain (the chains are independent between two swap rounds); force: NULL or one byte per step
 * (-1 / 0 / 1); per-step records (any may be NULL): log alpha, log u, scale, decided-on-a-stale-likelihood, natural decision */
void orc_replica_run(orc_replica* r, int i0, int i1, const int8_t* force, double* logalpha, double* logu, double* scale,
                     int8_t* stale, int8_t* natural, const float* sync_w, int64_t sync_stride, const float* sync_eta, int64_t sync_eta_stride,
                     double* la_sync, double* scale_sync, double* lik_sync) {
    /* sync_w (or NULL): the other implementation's recorded pos_w rows (float32), the row MH step i wrote (trace row i + 1) at
     * sync_w + (i - i0) * sync_stride; sync_eta (or NULL): its eta after step i at sync_eta[(i - i0) * sync_eta_stride].  After
     * every ACCEPTED step the chain's state is set from them (orc_replica_set_state).  The chain's own trace rows keep its own
     * float64 proposal, so comparing them with the other side's rows measures ONE step from a common state.
     *
     * la_sync / scale_sync / lik_sync (or NULL; NaN on rejected steps): the accepted step RE-EVALUATED AT THE OTHER SIDE'S OWN
     * PROPOSAL -- the recorded row is the proposal it accepted, so log alpha = (lik(w') - lik) + (prior(w') - prior) + diff_prop
     * (REG:365-372) is formed here in float64 from exactly the inputs the other side had: the common state w, its proposal w' and
     * eta', this chain's cached likelihood / prior of w, and for a Langevin step first = -|w - sgd(w')|^2 / 2 step^2 with this
     * chain's float64 epoch of w', second = -|noise|^2 / 2.  What differs from the other side's recorded value is then the
     * arithmetic of ONE evaluation (forward passes, sums, one SGD epoch), not the position of the proposal: with tau^2 ~ 1e-4 a
     * proposal moved by half a float32 ulp already changes log alpha by 1e-4 .. 1e-2.  lik_sync = the likeh_list entry of that
     * step at the other side's proposal (REG:391 tempered / CLS:404 times adapttemp). */
    const int P = r->P;
    double* wtmp = sync_w ? (double*)malloc((size_t)2 * P * sizeof(double)) : NULL;
    double* w_before = wtmp ? wtmp + P : NULL;
    for (int i = i0; i < i1; ++i) {
        if (la_sync) la_sync[i - i0] = NAN;
        if (scale_sync) scale_sync[i - i0] = NAN;
        if (lik_sync) lik_sync[i - i0] = NAN;
        if (sync_w) memcpy(w_before, r->w, (size_t)P * sizeof(double));
        const int took = orc_replica_step(r, i, force ? (int)force[i - i0] : -1);
        if (logalpha) logalpha[i - i0] = r->last_logalpha;
        if (logu) logu[i - i0] = log(r->last_u);
        if (scale) scale[i - i0] = r->last_scale;
        if (stale) stale[i - i0] = (int8_t)r->last_stale;
        if (natural) natural[i - i0] = (int8_t)r->last_natural;
        if (took && sync_w) {
            const float* row = sync_w + (size_t)(i - i0) * (size_t)sync_stride;
            for (int j = 0; j < P; ++j) wtmp[j] = (double)row[j];
            const double lik0 = r->last_lik_cur, pri0 = r->last_prior_cur;
            orc_replica_set_state(r, wtmp, sync_eta ? (double)sync_eta[(size_t)(i - i0) * (size_t)sync_eta_stride] : NAN);
            if (r->task == TASK_REG) r->tau_pro = exp(r->eta);      /* the last proposed tau IS this accepted one (Q9 reads it at the switch) */
            if (la_sync || scale_sync) {
                double diff = 0.0, first = 0.0, second = 0.0;
                if (r->last_lg) {
                    double *noise = r->scratch, *w_pgd = noise + 3 * P, *sc = noise + 4 * P;
                    orc_langevin_gradient(r->task, r->I, r->H, r->O, r->train, r->Ntr, r->ncols, wtmp, r->lr, w_pgd, sc);
                    double a = 0.0, b = 0.0;
                    for (int j = 0; j < P; ++j) { const double dc = w_before[j] - w_pgd[j]; a += dc * dc; b += noise[j] * noise[j]; }
                    first = -0.5 * a / (r->step_w * r->step_w); second = -0.5 * b;
                    diff = (first - second) / r->adapttemp;
                }
                if (la_sync) la_sync[i - i0] = (r->likelihood - lik0) + (r->prior_current - pri0) + diff;
                if (scale_sync) scale_sync[i - i0] = fabs(r->likelihood) + fabs(lik0) + fabs(r->prior_current) + fabs(pri0) +
                                                     (fabs(first) + fabs(second)) / r->adapttemp;
            }
            if (lik_sync) lik_sync[i - i0] = (r->task == TASK_REG) ? r->likelihood : r->likelihood * r->adapttemp;
        }
    }
    free(wtmp);
}

int orc_replica_struct_bytes(void) { return (int)sizeof(orc_replica); }
