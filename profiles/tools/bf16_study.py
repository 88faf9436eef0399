#!/usr/bin/env python3
"""fp32 vs bf16 tolerance study of BASELINE config 5 (north_star: "MFMA replica-batched GEMM path, fp32 vs bf16 tolerance study";
SURVEY 8d: "report log-lik abs error distribution and accept-decision flip rate vs fp32").

    python profiles/tools/bf16_study.py profiles/r04_bf16_study.json [S] [R]     # on the GPU box; ~ 5 - 8 minutes (the float64 oracle
                                                                                 # of this net needs ~ 50 ms per MH step per chain)

The net of bench.py's `synthetic512` workload (FNN 32-512-1, P = 17 409, 1024 / 256 rows, SURVEY 8d recipe), R chains (default
128 = one GPU's share of config 5), Langevin-gradient proposals p = 0.5, S samples per chain (default 201), swap every 50 steps,
run three times from the same start on the same random tape with the forward GEMM in each of the library's modes:

    split   (forward_bf16 = 0, the default)  fp32 operands split into three bf16 terms, six partial products: fp32 accuracy
    exact   (forward_bf16 = 2)               the fp32 matrix instruction (v_mfma_f32_32x32x2_f32)
    bf16    (forward_bf16 = 1)               operands ROUNDED to bf16, fp32 accumulation (v_mfma_f32_32x32x16_bf16): the study mode

Each run is followed by the float64 C oracle with the device's decisions and state imposed (tests/parity.py: follow_device_run):
  * |d loglik|   device's recorded log-likelihood of every ACCEPTED step against the oracle's at the device's own (w', eta')
                 -- identical inputs, so this is the error of the forward pass + likelihood sum alone;
  * |d log alpha| the same for the MH statistic of the accepted steps, and over ALL steps (each side forming its own proposal);
  * flips        MH decisions the float64 oracle would have taken differently from the common state, per 10^4 decisions, and how
                 many of them lie outside the fp32 coin-flip bound;
  * posterior    per-weight posterior mean of the kept half of the run against the `exact` run's, in units of that run's
                 posterior sd per chain and weight (chains part ways after the first flipped decision, so over a short run this
                 measures how far a flipped decision carries, not a bias);
  * samples/s    whole runs, timed as bench.py times them (K = 3 after one warm-up).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import numpy as np  # noqa: E402
import parity  # noqa: E402
from parity import orc  # noqa: E402
import ptnn_oracle_c as orc_c  # noqa: E402

_cli = __name__ == "__main__"
OUT = sys.argv[1] if _cli and len(sys.argv) > 1 else "/dev/null"
S = int(sys.argv[2]) if _cli and len(sys.argv) > 2 else 201
R = int(sys.argv[3]) if _cli and len(sys.argv) > 3 else 128
SI, SEED, TOPO = 50, 1, (32, 512, 1)
MODES = [("split", 0), ("exact", 2), ("bf16", 1)]


def dist(a):
    a = np.asarray(a, dtype=np.float64)
    a = a[np.isfinite(a)]
    if a.size == 0:
        return dict(n=0)
    return dict(n=int(a.size), median=float(np.median(a)), p90=float(np.percentile(a, 90)), p99=float(np.percentile(a, 99)), max=float(a.max()))


def run_study(R=R, S=S, SI=SI, modes=MODES, threads=16, timed_runs=3):
    train, test = parity.synthetic_regression(1280, 1024, 32, 512, seed=5)
    train, test = (np.asarray(a, dtype=np.float32).astype(np.float64) for a in (train, test))      # both sides read the device's values
    P = orc.num_param(TOPO)
    out = dict(workload=f"FNN 32-512-1 (P = {P}), {R} chains, Langevin p = 0.5 lr 0.1, maxtemp 2, S = {S} samples per chain, swap every {SI} steps, "
                        f"seed {SEED}, initial weights 0.3 x N(0, 1) (bench.py's scale for wide nets)", modes={})
    post = {}
    for name, fb in modes:
        t0 = time.time()
        pt = orc.PTOracle(orc.TASK_REG, TOPO, train, test, R, 2, R * S, SI, use_lg=True, l_prob=0.5, lr=0.1, seed=SEED)
        w0 = (0.3 * np.stack([rep.w for rep in pt.replicas])).astype(np.float32)
        orc_c.adopt(pt, w0=w0.astype(np.float64))
        s = parity.make_sampler(orc.TASK_REG, TOPO, train, test, R_local=R, R_global=R, first=0, S=S, si=SI, use_lg=True, lr=0.1, seed=SEED,
                                forward_bf16=fb)
        temps = np.array(pt.temperatures, dtype=np.float32)

        def whole_run():
            s.set_state(w0, temps)
            s.run(-1)
            s.sync()
        whole_run()
        t1 = time.perf_counter()
        for _ in range(timed_runs):
            whole_run()
        rate = R * (S - 1) * timed_runs / (time.perf_counter() - t1)
        tr = s.traces()
        info = s.describe()
        arrays = {}
        rep = parity.follow_device_run(s, tr, pt, f"{name} ", threads=threads, arrays=arrays, study=True)
        s.close()
        dec = int(rep["steps"])
        flips = int(rep["forced_mh"])
        b = S // 2
        pw = np.asarray(tr["pos_w"], dtype=np.float64)[:, b:, :]
        post[name] = (pw.mean(axis=1), pw.std(axis=1))
        out["modes"][name] = dict(
            forward_bf16=fb, kernel=info["kernel"], schedule=info["schedule"], samples_per_s=rate,
            decisions=dec, accepted=int(rep["accepted"]), flips=flips, flips_per_1e4=1e4 * flips / dec, flips_outside_coin_flip_bound=int(rep["flips_outside_bound"]),
            swap_pairs=int(rep["swap_pairs"]), swap_flips=int(rep["forced_swap_pairs"]),
            abs_err_loglik_identical_inputs=dist(arrays["err_lik_ident"]),
            rel_err_loglik_identical_inputs=dist(arrays["err_lik_ident"] / np.maximum(np.abs(arrays["lik_ident"]), 1e-300)),
            abs_err_logalpha_identical_inputs=dist(arrays["err_ident"]),
            abs_err_logalpha_all_steps=dist(arrays["err"]),
            logalpha_scale=dist(arrays["scale"]),
            oracle_seconds=round(time.time() - t0, 1))
        print(name, json.dumps(out["modes"][name]), flush=True)
    m0, s0 = post["exact"]
    for name, _ in modes:
        if name == "exact":
            continue
        m, _sd = post[name]
        z = np.abs(m - m0) / np.maximum(s0, 1e-12)
        moved = s0 > 0                                                      # a chain that accepted nothing in the kept half has sd 0
        out["modes"][name]["posterior_mean_shift_vs_exact_in_sd"] = dist(z[moved])
        out["modes"][name]["chains_identical_to_exact"] = int((np.abs(m - m0).max(axis=1) == 0).sum())
    return out


def main():
    out = run_study()
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("written", OUT)


if __name__ == "__main__":
    main()
