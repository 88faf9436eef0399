#!/usr/bin/env python3
"""bench.py -- MCMC samples/sec (all replicas) + swap-accept rate on the BASELINE.json workload.

Workload (N = 1): Sunspot one-step-ahead regression, FNN [4,5,1], 64 replicas, Langevin-gradient proposals with
probability 0.5 (lr 0.1), maxtemp 2, swap interval 100 -- the configuration BASELINE.json's metric is quoted on.
A bench "step" is one swap interval: 100 MH steps of every replica (one launch of the fused segment kernel) plus the
swap round.  Inputs (data set, weights, traces) are resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling, 64 replicas per GPU on a ladder of 64 N
temperatures, one RCCL all-gather + point-to-point row exchange per swap round (distributed.py).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TOPO = (4, 5, 1)
R_PER_GPU = 64
SWAP_INTERVAL = 100
L_PROB, LR, MAXTEMP = 0.5, 0.1, 2
SEED = 1
P = TOPO[0] * TOPO[1] + TOPO[1] * TOPO[2] + TOPO[1] + TOPO[2]
# mandatory HBM traffic of one MH step of one replica = the trace row the result files require,
# 4 (P + 7) bytes (pos_w row + likeh + 2 rmse + 2 acc + accept count); a swap round adds 4 (P + 2) per replica
B_STEP = 4 * (P + 7)
B_SWAP = 4 * (P + 2)
# flop per step (Sunspot [4,5,1]): F_RW = 42 811, F_LGextra = 105 082
F_STEP = 42811 + L_PROB * 105082
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
# Chain burn-in before warm-up and timing (Langevin workloads).  The speculative schedules run faster the fewer steps a chain
# accepts, and acceptance falls as the chains settle (Sunspot: 5 % over the first 1000 steps, 1.2 % around step 10 000), so the
# throughput of an interval depends on where in the chain it lies.  The timed region therefore always starts after the
# reference's own burn-in of a standard run (burn_in 0.5 x 10 000 samples per chain, REG:949,1004): whatever --steps and
# --warmup are, the bench samples the phase whose samples the reference keeps.  DESIGN.md 8 lists the throughput of every
# phase, start-up included.
BURN_IN_INTERVALS = 50
VALU_PEAK_TFLOPS = 157.3


def load_sunspot():
    path = os.path.join(ROOT, "tests", "golden", "datasets.npz")
    if os.path.exists(path):
        d = np.load(path)
        return d["sunspot_train"], d["sunspot_test"], "sunspot series shipped as tests/golden/datasets.npz (298/198 rows x 4 lags)"
    # same shape, synthetic: a noisy quasi-periodic series in [0,1] embedded with window 5 / stride 2
    rng = np.random.default_rng(0)
    t = np.arange(1000)
    s = 0.5 + 0.35 * np.sin(2 * np.pi * t / 44.0) * np.sin(2 * np.pi * t / 400.0) + 0.05 * rng.standard_normal(1000)
    s = (s - s.min()) / (s.max() - s.min())
    rows = np.stack([s[2 * k:2 * k + 5] for k in range(496)])
    return rows[:298], rows[298:], "synthetic sunspot-shaped series (298/198 rows x 4 lags)"


def switch_step(S):
    pt = S * 0.6
    return int(pt) if pt == int(pt) else -1


def make_sampler(train, test, R_local, R_global, first, S, device, use_lg=True, schedule=0, waves=0, groups=0):
    import ptnn_amd
    from ptnn_amd import _lib, ladder, philox
    s = _lib.Sampler(device_id=device, task=_lib.TASK_REG, n_in=TOPO[0], n_hidden=TOPO[1], n_out=TOPO[2],
                     n_replicas_local=R_local, n_replicas_global=R_global, first_global_replica=first, n_samples=S,
                     swap_interval=SWAP_INTERVAL, pt_switch_step=switch_step(S), use_langevin=int(use_lg),
                     waves_per_replica=waves, schedule=schedule, groups_per_replica=groups, l_prob=L_PROB, learn_rate=LR, step_w=0.025, step_eta=0.2, sigma_squared=25.0,
                     nu_1=0.0, nu_2=0.0, seed=SEED)
    s.set_data(train, test)
    T = ladder.temperatures(R_global, MAXTEMP)[first:first + R_local]
    w0 = np.stack([philox.initial_weights(SEED, first + r, P) for r in range(R_local)])
    s.set_state(w0, T)
    return s


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_chain(args):
    gid, n_steps, train, test, T = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ptnn_oracle as orc
    tape = orc.PhiloxTape(SEED)
    rep = orc.Replica(orc.TASK_REG, TOPO, train, test, tape.w_init(gid, P), T, 10 * n_steps, True, L_PROB, LR, tape, gid,
                      faithful=True)
    t0 = time.perf_counter()
    for i in range(n_steps):
        rep.step(i)
    return time.perf_counter() - t0


def usable_cores():
    """Host cores this process may actually use: the affinity mask and the cgroup CPU quota, not the machine's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except Exception:
            continue
    return n


def cpu_baseline(train, test, n_steps=40):
    """The oracle (float64 numpy restatement, per-row loops like the reference: faithful=True) on the host cores:
    the same 64-replica Langevin workload, n_steps MH steps per replica, one process per core."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ptnn_oracle as orc
    cores = max(1, min(usable_cores(), R_PER_GPU))
    T = orc.temperature_ladder(R_PER_GPU, MAXTEMP)
    jobs = [(g, n_steps, train, test, T[g]) for g in range(R_PER_GPU)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        busy = pool.map(_cpu_chain, jobs, chunksize=1)
    wall = time.perf_counter() - t0
    return {"value": R_PER_GPU * n_steps / wall, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{R_PER_GPU} replicas x {n_steps} MH steps (Langevin p=0.5), oracle faithful mode, "
                      f"{cores} processes, {sum(busy):.1f} s of CPU work, no swap rounds"}


# ------------------------------------------------------------------------------------------------ config 5 shape
def bench_synthetic512(a):
    """BASELINE config 5 shape on one GPU: FNN 32-512-1 (P = 17 409), 1024 train + 256 test rows (SURVEY 8d recipe),
    128 replicas, random-walk proposals: a step is dominated by the per-replica forward GEMM 1280 x 32 x 512 on MFMA."""
    import ptnn_amd
    from ptnn_amd import _lib, ladder, philox
    rng = np.random.default_rng(5)
    I, H, R, si = 32, 512, 128, 20
    Pw = I * H + H + H + 1
    X = rng.uniform(0, 1, (1280, I))
    wt = np.concatenate([rng.standard_normal(I * H) / np.sqrt(I), rng.standard_normal(H) / np.sqrt(H),
                         rng.standard_normal(H) / np.sqrt(I), rng.standard_normal(1) / np.sqrt(H)])
    sig = lambda z: 1.0 / (1.0 + np.exp(-z))
    y = np.clip(sig(sig(X @ wt[:I * H].reshape(I, H) - wt[I * H + H:I * H + 2 * H]) @ wt[I * H:I * H + H] - wt[-1])
                + rng.normal(0, 0.02, 1280), 0, 1)
    data = np.hstack([X, y[:, None]])
    K, W = a.steps, a.warmup
    S = (W + K + 1) * si + 2
    s = _lib.Sampler(device_id=int(os.environ.get("LOCAL_RANK", "0")), task=_lib.TASK_REG, n_in=I, n_hidden=H, n_out=1,
                     n_replicas_local=R, n_replicas_global=R, first_global_replica=0, n_samples=S, swap_interval=si,
                     pt_switch_step=switch_step(S), use_langevin=0, l_prob=0.5, learn_rate=0.1, step_w=0.025, step_eta=0.2,
                     sigma_squared=25.0, seed=SEED, forward_bf16=int(a.bf16), trace_capacity=4 * si)
    s.set_data(data[:1024], data[1024:])
    s.set_state(0.3 * np.stack([philox.initial_weights(SEED, r, Pw) for r in range(R)]), ladder.temperatures(R, MAXTEMP))
    drained = 0

    def advance(n_intervals):
        nonlocal drained
        for _ in range(n_intervals):                        # the trace ring holds 4 intervals: drain as we go
            s.run(si)
            hi = s.steps_done() + 1
            s.traces(drained, hi - drained, pos_w=False)
            drained = hi
    s.run(1)
    advance(W)
    s.sync()
    s.kernel_time(reset=True)
    t0 = time.perf_counter()
    advance(K)
    s.sync()
    dt = time.perf_counter() - t0
    launches, kms = s.kernel_time()
    value = R * K * si / dt
    flops_step = 2.0 * 1280 * I * H                          # the forward GEMM of one MH step of one replica
    avg_launch_s = kms / max(launches, 1) * 1e-3
    achieved = R * si * flops_step / avg_launch_s / 1e12
    peak = 2500.0 if a.bf16 else 157.3                      # MI355X_MICROARCH.md: dense bf16 MFMA / fp32 MFMA (= vector) peak
    print(json.dumps({
        "metric": "MCMC samples/sec (all replicas); synthetic 32-512-1, 128 replicas (BASELINE config 5 shape, one GPU)",
        "value": value, "unit": "samples/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.bf16 else "f32",
        "data": "synthetic (SURVEY.md 8d config-5 recipe, rng 5)",
        "config": {"workload": f"FNN 32-512-1, 1024/256 rows, {R} replicas, random-walk, swap every {si} steps; "
                               f"1 bench step = 1 swap interval; forward GEMM on MFMA ({'bf16' if a.bf16 else 'fp32'})"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                     "traffic": None, "kernel": "ptnn::segment_wide_kernel<0,32,1>", "avg_launch_ms": avg_launch_s * 1e3,
                     "launches": launches, "algorithmic_flops_per_launch": R * si * flops_step,
                     "note": "flops of the forward GEMM only; the launch also generates 17 409 normals, streams five 70 KB "
                             "vectors and writes a 70 KB trace row per step and replica"}}), flush=True)
    s.close()


# ------------------------------------------------------------------------------------------------ BASELINE configs 2-4
# name: (task, topology, data set, replicas, Langevin, lr, maxtemp, swap interval, description)
OTHER_CONFIGS = {
    "iris16": (1, (4, 12, 3), "iris", 16, False, 0.01, 10, 100, "BASELINE config 2: Iris FNN 4-12-3, 16 replicas, random-walk"),
    "mackey64": (0, (4, 10, 1), "mackey", 64, True, 0.1, 2, 100, "BASELINE config 3: Mackey-Glass FNN 4-10-1, 64 replicas, Langevin p=0.5"),
    "ionosphere256": (1, (34, 50, 2), "ions", 256, False, 0.01, 10, 100,
                      "BASELINE config 4 shape on one GPU: Ionosphere FNN 34-50-2, 256 replicas, random-walk, swap every 100 steps"),
}


def pmc_traffic(workload):
    """HBM bytes per launch of the dominant kernel of a workload, from the PMC passes committed under profiles/ (rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE, separate runs of the same command, gfx950 FETCH_SIZE x2 correction applied), or None."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", f"current_pmc_{workload}.json")))
        best = max((e for k, e in pmc["kernels"].items() if "segment" in k), key=lambda e: e.get("hbm_bytes_per_launch", 0.0))
        return best.get("hbm_bytes_per_launch")
    except Exception:
        return None


def bench_other_config(a):
    """Single-GPU measurement of the other BASELINE configs (parity-test cases; same JSON shape as the headline)."""
    import ptnn_amd
    from ptnn_amd import _lib, ladder, philox
    task, topo, dname, R, use_lg, lr, maxtemp, si, desc = OTHER_CONFIGS[a.workload]
    d = np.load(os.path.join(ROOT, "tests", "golden", "datasets.npz"))
    train, test = d[dname + "_train"], d[dname + "_test"]
    Pw = topo[0] * topo[1] + topo[1] * topo[2] + topo[1] + topo[2]
    K, W = a.steps, a.warmup
    B = BURN_IN_INTERVALS if (use_lg and not a.no_burn_in) else 0
    S = (B + W + K + 1) * si + 2
    s = _lib.Sampler(device_id=int(os.environ.get("LOCAL_RANK", "0")), task=task, n_in=topo[0], n_hidden=topo[1], n_out=topo[2],
                     n_replicas_local=R, n_replicas_global=R, first_global_replica=0, n_samples=S, swap_interval=si,
                     pt_switch_step=switch_step(S), use_langevin=int(use_lg), waves_per_replica=a.waves, schedule=a.schedule,
                     groups_per_replica=a.groups, l_prob=0.5, learn_rate=lr, step_w=0.025, step_eta=0.2, sigma_squared=25.0,
                     seed=SEED, trace_capacity=8 * si if Pw > 500 else 0)
    s.set_data(train, test)
    s.set_state(np.stack([philox.initial_weights(SEED, r, Pw) for r in range(R)]), ladder.temperatures(R, maxtemp))
    drained = 0

    def advance(n_steps):
        nonlocal drained
        left = n_steps
        while left > 0:                                      # drain the trace ring when there is one
            n = min(left, 4 * si)
            s.run(n)
            left -= n
            if Pw > 500:
                hi = s.steps_done() + 1
                s.traces(drained, hi - drained, pos_w=False)
                drained = hi
    advance((B + W) * si + (1 if task == 0 else 0))         # REG hands off after step k*si, CLS after step k*si - 1
    s.sync()
    s.kernel_time(reset=True)
    nsw0, tot0, _ = s.swap_stats()
    t0 = time.perf_counter()
    advance(K * si)
    s.sync()
    dt = time.perf_counter() - t0
    launches, kms = s.kernel_time()
    nsw1, tot1, _ = s.swap_stats()
    value = R * K * si / dt
    avg_launch_s = kms / max(launches, 1) * 1e-3
    bytes_per_launch = R * (K * si / max(launches, 1)) * 4 * (Pw + 7) + R * 4 * (Pw + 2)
    achieved = bytes_per_launch / avg_launch_s / 1e9
    print(json.dumps({
        "metric": "MCMC samples/sec (all replicas) + swap-accept rate; " + desc, "value": value, "unit": "samples/s", "n_gpus": 1,
        "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": f"{dname} data set shipped as tests/golden/datasets.npz ({train.shape[0]}/{test.shape[0]} rows)",
        "config": {"workload": desc + f"; 1 bench step = 1 swap interval of {si} MH steps", "replicas": R},
        "swap_accept_pct": 100.0 * (nsw1 - nsw0) / max(tot1 - tot0, 1),
        "mh_accept_pct": float(100.0 * np.mean(s.state()["num_accepted"]) / max(s.steps_done(), 1)),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(a.workload), "avg_launch_ms": avg_launch_s * 1e3, "launches": launches,
                     "algorithmic_bytes_per_launch": bytes_per_launch}}), flush=True)
    s.close()


def dependent_chain(s, W, K, si, slots=16):
    """What bounds the wall time of an interval (after the timed region, from the accept counters of the trace): an accepted
    step forces a new speculative round, and the swap barrier waits for the replica with the most of them."""
    acc = s.traces(0, s.steps_done() + 1, pos_w=False)["accept"].astype(np.int64)   # acc[r, i+1] = accepted before step i
    flags = np.diff(acc, axis=1)[:, 1:]
    a0 = W * si + 1
    per_acc, per_rounds = [], []
    for it in range(K):
        f = flags[:, a0 + it * si: a0 + (it + 1) * si]
        per_acc.append(f.sum(axis=1))
        rr = []
        for row in f:
            pos, rounds = 0, 0
            while pos < row.shape[0]:
                hit = np.flatnonzero(row[pos:pos + slots])
                pos += (hit[0] + 1) if hit.size else slots
                rounds += 1
            rr.append(rounds)
        per_rounds.append(rr)
    per_acc, per_rounds = np.array(per_acc), np.array(per_rounds)
    # floor of an interval: every accepted step of its slowest replica costs one sequential SGD epoch (298 rows x 143 cycles
    # per row measured with in-kernel stamps, DESIGN.md 4, at the 2.4 GHz shader clock of the stamp runs)
    epoch_ms = 298 * 143 / 2.4e9 * 1e3
    floor_ms = float(per_acc.max(axis=1).mean()) * epoch_ms
    return {"slots_per_round": slots, "sgd_epoch_ms": epoch_ms, "chain_floor_ms_per_interval": floor_ms,
            "accepted_steps_per_interval_mean": float(per_acc.mean()),
            "accepted_steps_per_interval_max_over_replicas_mean": float(per_acc.max(axis=1).mean()),
            "rounds_per_interval_mean": float(per_rounds.mean()),
            "rounds_per_interval_max_over_replicas_mean": float(per_rounds.max(axis=1).mean()),
            "note": "an interval ends when its slowest replica does (synchronous swap barrier, REG:730-752); every accepted "
                    "Langevin step costs one more sequential SGD epoch whatever the number of speculative slots"}


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rw", action="store_true", help="random-walk proposals only (extra data point)")
    ap.add_argument("--no-whole-run", action="store_true", help="skip the extra whole-run leg (from_chain_start); used under rocprofv3 so "
                    "that its per-kernel averages cover the launches of the measured sampler only")
    ap.add_argument("--no-burn-in", action="store_true", help="start warm-up and timing at chain step 0 (start-up transient included)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="sunspot64", choices=["sunspot64", "synthetic512", "iris16", "mackey64", "ionosphere256"],
                    help="sunspot64 = the BASELINE metric (default); synthetic512 = BASELINE config 5 shape (FNN 32-512-1, "
                         "1024/256 rows, 128 replicas per GPU, random-walk): the MFMA forward pass, roofline bound 'mfma'")
    ap.add_argument("--bf16", action="store_true", help="synthetic512: forward GEMM operands in bf16")
    ap.add_argument("--schedule", type=int, default=0, help="0 auto, 1 cooperative, 2 speculative")
    ap.add_argument("--waves", type=int, default=0, help="wavefronts per work-group (0 = auto)")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded-ladder driver even at world size 1 (rehearsal)")
    ap.add_argument("--groups", type=int, default=0, help="work-groups (CUs) per replica, speculative schedule (0 = auto)")
    a = ap.parse_args()
    K, W, N = a.steps, a.warmup, a.gpus
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != N:
        if world == 1 and N > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        N = world
    if a.workload == "synthetic512":
        return bench_synthetic512(a)
    if a.workload in OTHER_CONFIGS:
        return bench_other_config(a)
    train, test, data_desc = load_sunspot()
    si = SWAP_INTERVAL
    use_lg = not a.rw
    B = 0 if a.no_burn_in else BURN_IN_INTERVALS               # chain burn-in before warm-up (see BURN_IN_INTERVALS)
    S = (B + W + K + 1) * si + 2
    R_global = R_PER_GPU * N

    cpu = None
    if N == 1 and rank == 0 and not a.no_cpu_baseline:
        cpu = cpu_baseline(train, test)      # before the first HIP call: the pool forks
    if N == 1 and not a.force_dist:
        s = make_sampler(train, test, R_PER_GPU, R_global, 0, S, local_rank, use_lg, a.schedule, a.waves, a.groups)
        s.run((B + W) * si + 1)      # REG hands off after step i = k*si (REG:427): start the timed region on an interval boundary
        s.sync()
        pre_launches, pre_ms = s.kernel_time(reset=True)     # burn-in + warm-up launches (rocprofv3 --stats averages them in)
        nsw0, tot0, _ = s.swap_stats()
        t0 = time.perf_counter()
        s.run(K * si)
        s.sync()
        dt = time.perf_counter() - t0
        launches, kms = s.kernel_time()
        nsw1, tot1, _ = s.swap_stats()
        accepted = s.state()["num_accepted"]
        steps_done = s.steps_done()
        chain = dependent_chain(s, B + W, K, si) if rank == 0 else None
        # beside it, untimed by the contract: a whole run of the reference's standard length from the chain start (start-up
        # transient and burn-in included), so that the line also says what a user's run_chains() sees end to end
        whole = None
        if rank == 0 and B > 0 and not a.no_whole_run:
            s2 = make_sampler(train, test, R_PER_GPU, R_global, 0, 100 * si + 2, local_rank, use_lg, a.schedule, a.waves, a.groups)
            s2.run(1)
            s2.sync()
            tw = time.perf_counter()
            s2.run(100 * si)
            s2.sync()
            whole = {"value": R_global * 100 * si / (time.perf_counter() - tw), "unit": "samples/s", "mh_steps": "1..%d" % (100 * si)}
            s2.close()
    else:
        chain = whole = None
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # PTNN_BENCH_REHEARSE=1: every rank on GPU 0, gloo transport -- lets a one-GPU box run the whole N > 1 code path
        # (RCCL refuses two ranks on one device); the figure it prints is not a measurement of anything
        rehearse = os.environ.get("PTNN_BENCH_REHEARSE", "0") == "1"
        if rehearse:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        from ptnn_amd import distributed as dm
        s = make_sampler(train, test, R_PER_GPU, R_global, rank * R_PER_GPU, S, local_rank, use_lg, a.schedule, a.waves, a.groups)
        lad = dm.ShardedLadder(dm.DeviceShard(s, local_rank), rank, N, dist)
        lad.run_intervals(B + W)
        s.sync()
        pre_launches, pre_ms = s.kernel_time(reset=True)
        nsw0, tot0, _ = s.swap_stats()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lad.run_intervals(K)
        s.sync()
        dist.barrier()
        torch.cuda.synchronize()
        dt_local = time.perf_counter() - t0
        t = torch.tensor([dt_local], device="cpu" if rehearse else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        launches, kms = s.kernel_time()
        nsw1, tot1, _ = s.swap_stats()
        accepted = s.state()["num_accepted"]
        steps_done = s.steps_done()

    if rank == 0:
        mh_steps = K * si                                       # per replica, inside the timed region
        value = R_global * mh_steps / dt
        avg_launch_s = (kms / max(launches, 1)) * 1e-3
        bytes_per_launch = R_PER_GPU * (mh_steps / max(launches, 1)) * B_STEP + R_PER_GPU * B_SWAP
        achieved = bytes_per_launch / avg_launch_s / 1e9 if launches else 0.0
        flops_per_launch = R_PER_GPU * (mh_steps / max(launches, 1)) * (F_STEP if use_lg else 42811)
        # HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE, separate runs of this same command, gfx950 FETCH_SIZE x2 correction applied)
        if a.schedule == 1:
            kname = "ptnn::segment_kernel<0,4,1>"
        elif a.schedule == 3 or (a.schedule == 0 and use_lg and a.waves == 0 and a.groups == 0):
            kname = "ptnn::segment_pack_kernel<0,4,1>"          # what schedule 0 resolves to for this workload
        else:
            kname = "ptnn::segment_spec_kernel<0,4,1>"
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "current_pmc.json")))
            for kn, e in pmc["kernels"].items():
                if kname.split("<")[0].split("::")[1] + "<" in kn and use_lg and N == 1 and a.waves == 0 and a.groups == 0:
                    traffic = e.get("hbm_bytes_per_launch")
        except Exception:
            pass
        out = {
            "metric": "MCMC samples/sec (all replicas) + swap-accept rate; Sunspot 64-replica FNN",
            "value": value, "unit": "samples/s", "n_gpus": N, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": data_desc,
            "config": {"workload": f"Sunspot FNN 4-5-1, {R_PER_GPU} replicas/GPU ({R_global} temperatures), "
                                   + ("Langevin p=0.5 lr=0.1" if use_lg else "random-walk") +
                                   f", maxtemp {MAXTEMP}, swap every {si} MH steps; 1 bench step = 1 swap interval",
                       "replicas": R_global, "mh_steps_per_bench_step": si, "proposals": "langevin" if use_lg else "rw",
                       "chain_burn_in_mh_steps": B * si, "first_timed_mh_step": (B + W) * si + 1, "last_timed_mh_step": (B + W + K) * si,
                       "schedule": a.schedule, "waves_per_replica": a.waves, "groups_per_replica": a.groups},
            "swap_accept_pct": 100.0 * (nsw1 - nsw0) / max(tot1 - tot0, 1),
            "mh_accept_pct": float(100.0 * np.mean(accepted) / max(steps_done, 1)),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kname, "avg_launch_ms": avg_launch_s * 1e3, "launches": launches,
                         "avg_launch_ms_all_launches": (kms + pre_ms) / max(launches + pre_launches, 1),
                         "all_launches": launches + pre_launches,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "valu_tflops": flops_per_launch / avg_launch_s / 1e12 if launches else 0.0,
                         "valu_frac": (flops_per_launch / avg_launch_s / 1e12) / VALU_PEAK_TFLOPS if launches else 0.0,
                         "traffic_source": "profiles/current_pmc.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per launch)",
                         "note": "instruction-issue bound by construction (sequential SGD rows, AI 627 flop/B); the HBM "
                                 "fraction is reported because BASELINE.json asks for it"},
        }
        if whole is not None:
            out["from_chain_start"] = whole
        if chain is not None:
            chain["frac_of_chain_floor"] = chain["chain_floor_ms_per_interval"] / (avg_launch_s * 1e3) if launches else None
            out["dependent_chain"] = chain
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
        print(json.dumps(out), flush=True)
    if N > 1 or a.force_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    s.close()


if __name__ == "__main__":
    main()
