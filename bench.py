#!/usr/bin/env python3
"""bench.py -- MCMC samples/sec (all replicas) + swap-accept rate on the BASELINE.json workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sunspot64|iris16|mackey64|ionosphere256|synthetic512]

Workload at N = 1 (default `sunspot64`, the configuration BASELINE.json's metric is quoted on): Sunspot one-step-ahead
regression, FNN [4,5,1], 64 replicas, Langevin-gradient proposals with probability 0.5 (lr 0.1), maxtemp 2, swap every 100 MH
steps, S = 10 000 samples per replica (the length of a standard run of the reference, REG:949-1004).

A bench STEP is ONE WHOLE RUN of that length FROM THE CHAIN START: chain start-up (eta0, initial likelihood and prior,
REG:253-285), the S - 1 MH steps of every replica and all int(S / swap_interval) swap rounds -- what the reference's own
figure means (NumSample / wall time of run_chains, REG:1019-1022).  The speculative schedules run faster where a chain
accepts fewer proposals and acceptance falls along a chain, so a window inside a run measures the window, not the sampler:
a whole run does not depend on where --steps / --warmup put it.  W warm-up runs, then K timed runs back to back (each restarts
the chains with ptnn_set_state, inside the timed region), bracketed by a barrier and a device synchronisation; the
maximum over ranks is the time.  Inputs (data set, initial weights) are resident in HBM before the timed region, the trace
rows the result files need are written to HBM inside it (sized for 288 GB: no host transfer in the timed region).

N > 1: one rank (process) per GPU.  Under torch.distributed.run the ranks are the launcher's; run bare (`python3 bench.py --gpus N`,
WORLD_SIZE unset) rank 0's parent starts the N ranks itself -- N fresh child processes of this script with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, before anything touches the GPU -- relays rank 0's JSON line and exits with the worst child's status.
Weak scaling by default (the same replicas per GPU on a ladder of N times as many temperatures); `--scaling strong` keeps the
workload's replica count and cuts it into N blocks (BASELINE config 4: `--workload ionosphere256 --scaling strong --gpus 8` =
256 replicas over 8 GPUs).  The swap rounds run inside libptnn over RCCL (ptnn_comm_init; torch.distributed is only the
rendezvous: it carries the RCCL unique id, the barrier and the max-over-ranks of the time); the line's `comm` object says what
the communicator itself saw (ranks, devices, bytes, rounds).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 1
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3        # fp32 vector (= fp32 MFMA) peak
MFMA_BF16_PEAK_TFLOPS = 2500.0


def n_param(topo):
    return topo[0] * topo[1] + topo[1] * topo[2] + topo[1] + topo[2]


def flops_per_step(topo, ntr, nte, p_lg):
    """E[F_step] = F_RW + p F_LGextra."""
    I, H, O = topo
    P = n_param(topo)
    f_fwd = 2 * (I * H + H * O) + 5 * (H + O)
    f_bwd = 2 * I * H + 4 * H * O + 6 * H + 6 * O
    f_rw = (ntr + nte) * f_fwd + 6 * O * (ntr + nte) + 5 * P
    f_lg = 2 * ntr * (f_fwd + f_bwd) + 6 * P
    return f_rw + p_lg * f_lg


# name: task, topology, data set, replicas per GPU, Langevin, lr, maxtemp, swap interval, samples per replica, description
WORKLOADS = {
    "sunspot64": dict(task=0, topo=(4, 5, 1), data="sunspot", R=64, lg=True, lr=0.1, maxtemp=2, si=100, S=10000,
                      desc="Sunspot FNN 4-5-1, 64 replicas/GPU, Langevin p=0.5 lr=0.1, maxtemp 2 (the BASELINE metric)"),
    "iris16": dict(task=1, topo=(4, 12, 3), data="iris", R=16, lg=False, lr=0.01, maxtemp=10, si=100, S=10000,
                   desc="BASELINE config 2: Iris FNN 4-12-3, 16 replicas, random-walk, maxtemp 10"),
    "mackey64": dict(task=0, topo=(4, 10, 1), data="mackey", R=64, lg=True, lr=0.1, maxtemp=2, si=100, S=10000,
                     desc="BASELINE config 3: Mackey-Glass FNN 4-10-1, 64 replicas, Langevin p=0.5 lr=0.1, maxtemp 2"),
    "ionosphere256": dict(task=1, topo=(34, 50, 2), data="ions", R=256, lg=False, lr=0.01, maxtemp=10, si=100, S=10000,
                          desc="BASELINE config 4 shape on one GPU: Ionosphere FNN 34-50-2, 256 replicas, random-walk, maxtemp 10"),
    "synthetic512": dict(task=0, topo=(32, 512, 1), data="synthetic512", R=128, lg=True, lr=0.1, maxtemp=2, si=100, S=1001,
                         desc="BASELINE config 5 shape on one GPU: synthetic FNN 32-512-1 (SURVEY 8d recipe), 128 replicas/GPU, "
                              "Langevin p=0.5 lr=0.1, maxtemp 2; S = 1001 samples per replica (a tenth of a standard run)"),
}


def load_data(name):
    if name == "synthetic512":
        rng = np.random.default_rng(5)
        I, H = 32, 512
        X = rng.uniform(0, 1, (1280, I))
        wt = np.concatenate([rng.standard_normal(I * H) / np.sqrt(I), rng.standard_normal(H) / np.sqrt(H),
                             rng.standard_normal(H) / np.sqrt(I), rng.standard_normal(1) / np.sqrt(H)])
        sig = lambda z: 1.0 / (1.0 + np.exp(-z))                                                              # noqa: E731
        y = np.clip(sig(sig(X @ wt[:I * H].reshape(I, H) - wt[I * H + H:I * H + 2 * H]) @ wt[I * H:I * H + H] - wt[-1])
                    + rng.normal(0, 0.02, 1280), 0, 1)
        data = np.hstack([X, y[:, None]])
        return data[:1024], data[1024:], "synthetic (SURVEY.md 8d config-5 recipe, rng 5: 1024/256 rows x 32 inputs)"
    d = np.load(os.path.join(ROOT, "tests", "golden", "datasets.npz"))
    tr, te = d[name + "_train"], d[name + "_test"]
    return tr, te, f"{name} data set shipped as tests/golden/datasets.npz ({tr.shape[0]}/{te.shape[0]} rows)"


def switch_step(S):
    pt = S * 0.6
    return int(pt) if pt == int(pt) else -1


class Ladder:
    """The sampler of one rank (one GPU): a contiguous block of the temperature ladder."""

    def __init__(self, wl, a, train, test, rank, world, device, shared_noise=None, shared_device=0):
        import ptnn_amd  # noqa: F401
        from ptnn_amd import _lib, ladder, philox
        topo, R = wl["topo"], wl["R"]
        self.P = n_param(topo)
        self.R_local, self.R_global, self.S = R, R * world, wl["S"]
        first = rank * R
        self.s = _lib.Sampler(device_id=device, task=wl["task"], n_in=topo[0], n_hidden=topo[1], n_out=topo[2],
                              n_replicas_local=R, n_replicas_global=self.R_global, first_global_replica=first, n_samples=self.S,
                              swap_interval=wl["si"], pt_switch_step=switch_step(self.S), use_langevin=int(wl["lg"]),
                              waves_per_replica=a.waves, schedule=a.schedule, groups_per_replica=a.groups, l_prob=0.5,
                              learn_rate=wl["lr"], step_w=0.025, step_eta=0.2, sigma_squared=25.0, nu_1=0.0, nu_2=0.0, seed=SEED,
                              forward_bf16=int(a.bf16), shared_noise=int(a.shared_noise if shared_noise is None else shared_noise),
                              shared_device=int(shared_device))
        self.s.set_data(train, test)
        scale = 0.3 if topo[1] > 64 else 1.0      # wide nets: N(0,1) weights saturate every hidden unit of a 512-unit layer
        self.w0 = scale * np.stack([philox.initial_weights(SEED, first + r, self.P) for r in range(R)])
        self.T = ladder.temperatures(self.R_global, wl["maxtemp"])[first:first + R]

    def whole_run(self, split=None):
        """One run of the reference's length from the chain start.  `split`: also return the host time at which MH step
        `split` was reached (one more device synchronisation), to report the throughput of the kept half on its own."""
        s = self.s
        s.set_state(self.w0, self.T)
        t_mid = None
        if split:
            s.run(split)
            s.sync()
            t_mid = time.perf_counter()
        s.run(-1)
        s.sync()
        return t_mid


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_chain(args):
    gid, n_steps, train, test, T, topo = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ptnn_oracle as orc
    tape = orc.PhiloxTape(SEED)
    rep = orc.Replica(orc.TASK_REG, topo, train, test, tape.w_init(gid, n_param(topo)), T, 10 * n_steps, True, 0.5, 0.1, tape, gid,
                      faithful=True)
    t0 = time.perf_counter()
    for i in range(n_steps):
        rep.step(i)
    return time.perf_counter() - t0, rep.posted_L()


def _cpu_chain_c(args):
    """The same chain in the C restatement (oracle/ptnn_oracle_c.c): MH steps 0 .. n_steps - 1 of one replica."""
    gid, n_steps, train, test, T, topo = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ptnn_oracle as orc
    import ptnn_oracle_c as orc_c
    tape = orc.PhiloxTape(SEED)
    rep = orc_c.CReplica(orc.TASK_REG, topo, train, test, tape.w_init(gid, n_param(topo)), T, n_steps + 1, True, 0.5, 0.1, SEED, gid)
    t0 = time.perf_counter()
    rep.run(0, n_steps)
    return time.perf_counter() - t0, rep.posted_L()


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


# The port against the reference itself, measured where both can run (the survey container, 8 cores):
# Sunspot [4,5,1], one chain in-process, the reference takes 3.81 / 10.47 ms per random-walk / Langevin step, the faithful oracle
# 2.79 / 7.53 ms -- reference time = port time x 1.38, i.e. reference samples/s = port samples/s / 1.38 on the same cores.
PORT_OVER_REFERENCE = 1.38


def cpu_baseline(wl, train, test, n_steps=101):
    """The oracle (float64 numpy restatement, per-row loops like the reference: faithful=True) on the host cores: the same
    64-replica Langevin workload -- the first swap interval of the run: MH steps 0 .. 100 of every replica, one process per core
    (the reference forks one per chain, REG:709-712), then the swap round of that interval in the parent (REG:741-748).  A
    bounded sample: the CPU cost of a step does not depend on where in the chain it lies."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ptnn_oracle as orc
    R = wl["R"]
    cores = max(1, min(usable_cores(), R))
    T = orc.temperature_ladder(R, wl["maxtemp"])
    jobs = [(g, n_steps, train, test, T[g], wl["topo"]) for g in range(R)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_chain, jobs, chunksize=1)
    _, nsw = orc.swap_cascade([L for _, L in res], orc.PhiloxTape(SEED).swap_uniforms(0, R - 1))
    wall = time.perf_counter() - t0
    value = R * n_steps / wall
    # the plain-C restatement of the same oracle on the same cores, a whole run's worth of MH steps per replica (no swaps): what a compiled
    # CPU port of the reference does (about 100 x the interpreted reference) -- reported beside the numpy port, which is the one
    # BASELINE.md 3.2 relates to the reference
    c_port = None
    try:
        n_c = 100 * (n_steps - 1) + 1
        jobs_c = [(g, n_c, train, test, T[g], wl["topo"]) for g in range(R)]
        t1 = time.perf_counter()
        with ctx.Pool(cores) as pool:
            res_c = pool.map(_cpu_chain_c, jobs_c, chunksize=1)
        wall_c = time.perf_counter() - t1
        c_port = {"value": R * n_c / wall_c, "unit": "samples/s", "cores": cores, "kind": "port (plain C, oracle/ptnn_oracle_c.c)",
                  "sample": f"{R} replicas x MH steps 0..{n_c - 1}, {cores} processes, {sum(t for t, _ in res_c):.1f} s of CPU work"}
    except Exception as e:                                      # noqa: BLE001  (no compiler on the box: the numpy leg stands alone)
        c_port = {"error": repr(e)}
    return {"value": value, "unit": f"samples/s (MH steps 0..{n_steps - 1} of every replica: the first swap interval only)", "cores": cores, "kind": "port",
            "c_port": c_port,
            "port_over_reference": PORT_OVER_REFERENCE,
            "reference_equivalent": value / PORT_OVER_REFERENCE,
            "port_over_reference_source": "BASELINE.md 3.2 / DESIGN.md 8: the reference and the faithful oracle timed on identical work in "
                                          "the survey container (3.81 / 10.47 ms vs 2.79 / 7.53 ms per RW / Langevin step)",
            "sample": f"{R} replicas x MH steps 0..{n_steps - 1} (the first swap interval, Langevin p=0.5) + its swap round "
                      f"({nsw} of {R - 1} pairs swapped), oracle faithful mode, {cores} processes, {sum(t for t, _ in res):.1f} s of CPU work"}


# ------------------------------------------------------------------------------------------------ diagnostics
def dependent_chain(tr_accept, si, slots, epoch_ms):
    """What bounds the wall time of an interval of a speculative schedule (from the accept counters of the last run's trace):
    an accepted step forces a new round, and the swap barrier waits for the replica with the most of them."""
    acc = tr_accept.astype(np.int64)                        # acc[r, i+1] = accepted before step i
    flags = np.diff(acc, axis=1)[:, 1:]                     # flags[r, i] = step i accepted, i = 0 .. S-3
    n_int = flags.shape[1] // si
    per_acc, per_rounds = [], []
    for it in range(n_int):
        f = flags[:, 1 + it * si: 1 + (it + 1) * si]          # REG: interval `it` = MH steps it si + 1 .. (it + 1) si
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
    return {"slots_per_round": slots, "sgd_epoch_ms": epoch_ms,
            "sgd_epoch_source": "measured in this process: ptnn_time_sgd_epoch, 200 epochs back to back on one wavefront, in-kernel "
                                "constant-rate counter",
            "chain_floor_ms_per_interval": float(per_acc.max(axis=1).mean()) * epoch_ms,
            "accepted_steps_per_interval_mean": float(per_acc.mean()),
            "accepted_steps_per_interval_max_over_replicas_mean": float(per_acc.max(axis=1).mean()),
            "rounds_per_interval_mean": float(per_rounds.mean()),
            "rounds_per_interval_max_over_replicas_mean": float(per_rounds.max(axis=1).mean()),
            "note": "over the whole run; an interval ends when its slowest replica does (synchronous swap barrier, REG:730-752) "
                    "and every accepted Langevin step costs one more sequential SGD epoch whatever the number of speculative slots"}


def pmc_traffic(workload, kernel, steps_per_launch):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes of this same command, gfx950 FETCH_SIZE x2 correction applied; profiles/summarize.py) -- only
    when "a launch" meant the same number of MH steps there as in this run (a schedule that has since moved from one launch per
    swap interval to one per run must be profiled again)."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", f"current_pmc_{workload}.json")))
        then = pmc.get("mh_steps_per_launch") or (pmc.get("bench_kt") or {}).get("roofline", {}).get("mh_steps_per_launch")
        if then and abs(then - steps_per_launch) > 0.01 * steps_per_launch:
            return None, None
        stem = kernel.split("::")[-1].split("<")[0] + "<"
        for kn, e in pmc["kernels"].items():
            if stem in kn and "hbm_bytes_per_launch" in e:
                return e["hbm_bytes_per_launch"], f"{pmc.get('tag')}, commit {pmc.get('commit')}"
    except Exception:
        pass
    return None, None


# ------------------------------------------------------------------------------------------------ N > 1 without a launcher
def launch_plan(argv, n, master_port=None, environ=None, python=None, script=None):
    """What `python3 bench.py --gpus N` run bare starts: one command line + environment per rank (a pure function: the CPU test
    reads it).  Every child is a fresh process of this same script with the same arguments; RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT are what torch.distributed.run would have set."""
    env0 = dict(os.environ if environ is None else environ)
    if master_port is None:
        import socket
        with socket.socket() as sk:                          # a free port, asked of the kernel
            sk.bind(("127.0.0.1", 0))
            master_port = sk.getsockname()[1]
    plans = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(master_port), PTNN_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        plans.append(([python or sys.executable, script or os.path.abspath(__file__)] + list(argv), env))
    return plans


def self_launch(argv, n, plans=None):
    """Start the N ranks, pass rank 0's stdout through (its ONE JSON line), the other ranks' stdout to stderr; wait for all; the
    exit status is the worst child's.  A rank that dies takes the others with it after a grace period (they would wait for it in
    a collective until their own bounds expire)."""
    import subprocess
    import threading
    plans = launch_plan(argv, n) if plans is None else plans
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True) for cmd, env in plans]

    def pump(r, p):
        for line in p.stdout:
            mine = r == 0 and line.lstrip().startswith("{")      # rank 0's JSON line and nothing else goes to this process's stdout
            (sys.stdout if mine else sys.stderr).write(line if mine else f"[rank {r}] {line}")
            (sys.stdout if mine else sys.stderr).flush()
    threads = [threading.Thread(target=pump, args=(r, p), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    worst, t_fail = 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        for p in procs:
            rc = p.poll()
            if rc not in (None, 0) and t_fail is None:
                t_fail = time.time()
        if t_fail is not None and time.time() - t_fail > float(os.environ.get("PTNN_COMM_TIMEOUT_S", "120")) + 60.0:
            for p in procs:
                if p.poll() is None:
                    p.kill()                                 # these pids only
    for t in threads:
        t.join(timeout=5.0)
    for p in procs:
        rc = p.returncode
        worst = max(worst, abs(rc) if rc is not None else 1)
    return min(worst, 255)


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed whole runs")
    ap.add_argument("--warmup", type=int, default=10, help="untimed whole runs before them")
    ap.add_argument("--workload", default="sunspot64", choices=list(WORKLOADS))
    ap.add_argument("--rw", action="store_true", help="random-walk proposals only (extra data point)")
    ap.add_argument("--replicas", type=int, default=0, help="replicas per GPU instead of the workload's own count (extra data point)")
    ap.add_argument("--bf16", action="store_true", help="synthetic512: forward GEMM operands in bf16 (tolerance study mode)")
    ap.add_argument("--shared-noise", type=int, default=1, choices=[0, 1],
                    help="1 (default, the drop-in classes' default): every chain reads ONE noise tape, as the reference's forked chains "
                         "do (REG:709-712, SURVEY Q14); 0: independent Philox streams per chain")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra legs (kept-half split run, dependent-chain "
                    "analysis): under rocprofv3 every launch of the process then belongs to a warm-up or a timed run")
    ap.add_argument("--schedule", type=int, default=0, help="0 auto, 1 cooperative, 2 speculative, 3 packed")
    ap.add_argument("--waves", type=int, default=0, help="wavefronts per work-group (0 = auto)")
    ap.add_argument("--groups", type=int, default=0, help="work-groups (CUs) per replica, speculative schedule (0 = auto)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "gather", "boundary"],
                    help="N > 1: what a swap round moves between GPUs (ptnn_comm_set_mode)")
    ap.add_argument("--force-comm", action="store_true", help="N = 1: attach a one-rank communicator anyway, so that the whole N > 1 code "
                    "path (torch.distributed rendezvous next to RCCL inside libptnn, swap rounds through the communicator) runs on a "
                    "one-GPU box")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = the workload's replicas PER GPU (N times the ladder); strong = the workload's replicas in "
                         "total, cut into N blocks (R / N per GPU)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "host"],
                    help="N > 1: rccl = RCCL over xGMI inside libptnn; host = host-staged through gloo (rehearsal of the N > 1 code "
                         "path with every rank on GPU 0 of a one-GPU box; what it prints is not a measurement)")
    a = ap.parse_args()
    K, W, N = a.steps, a.warmup, a.gpus
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched_by_env = "WORLD_SIZE" in os.environ             # a launcher (torch.distributed.run, or this script itself) set the rank variables
    if world != N:
        if "WORLD_SIZE" not in os.environ and N > 1:
            # bare `python3 bench.py --gpus N`: this process becomes the launcher -- nothing has touched the GPU yet
            sys.exit(self_launch(sys.argv[1:], N))
        if world == 1 and N > 1:
            raise SystemExit(f"--gpus {N} but WORLD_SIZE=1: unset WORLD_SIZE (bench.py then starts its own ranks) or launch with "
                             f"torch.distributed.run --nproc-per-node {N}")
        N = world
    wl = dict(WORKLOADS[a.workload])
    if a.replicas:
        wl["desc"] = wl["desc"].replace(f"{wl['R']} replicas", f"{a.replicas} replicas")
        wl["R"] = a.replicas
    if a.scaling == "strong" and N > 1:
        if wl["R"] % N:
            raise SystemExit(f"--scaling strong: {wl['R']} replicas cannot be cut into {N} equal blocks")
        wl["desc"] += f" [strong scaling: {wl['R']} replicas in total, {wl['R'] // N} per GPU]"
        wl["R"] = wl["R"] // N
    if a.rw:
        wl["lg"] = False
        wl["desc"] = wl["desc"].replace("Langevin p=0.5 lr=0.1", "random-walk").replace("Langevin p=0.5", "random-walk")
    train, test, data_desc = load_data(wl["data"])
    S, si, R = wl["S"], wl["si"], wl["R"]

    cpu = None
    if N == 1 and rank == 0 and a.workload == "sunspot64" and not a.rw and not a.no_cpu_baseline:
        cpu = cpu_baseline(wl, train, test)      # before the first HIP call: the pool forks

    dist = None
    sharded = N > 1 or a.force_comm
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import ptnn_amd  # noqa: F401
        from ptnn_amd import distributed as dm0
        dm0.single_node_rccl_env()               # this process is the bench's own: loopback bootstrap, no verbs probe, dmabuf IPC
        import torch
        import torch.distributed as dist
        # rendezvous only: unique id, barrier, max of the times.  gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected
        # to ..."): the line this script prints must be the only thing there, so file descriptor 1 points at stderr while gloo comes up
        sys.stdout.flush()
        fd1 = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            os.dup2(fd1, 1)
            os.close(fd1)
        ngpu = torch.cuda.device_count()         # counting devices does not initialise the GPU
        device = local_rank % max(ngpu, 1)       # a one-GPU box rehearsing N ranks puts them all on GPU 0
        torch.cuda.set_device(device)
    else:
        ngpu = 1
        device = local_rank
    # ranks that share a GPU: libptnn then keeps to schedules whose work-groups never wait for each other
    shared_dev = int(sharded and N > max(ngpu, 1))
    lad = Ladder(wl, a, train, test, rank, N, device, shared_device=shared_dev)
    s = lad.s
    transport_used, transport_note = ("none", None)
    if sharded:
        import torch
        from ptnn_amd import _lib
        from ptnn_amd import distributed as dm
        mode = {"auto": 0, "gather": 1, "boundary": 2}[a.exchange]

        def all_ok(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        transport_used = a.transport
        if a.transport == "rccl" and shared_dev and not dm.injected_fault():
            # more ranks than GPUs (a one-GPU box rehearsing N ranks): RCCL refuses two ranks on one device, and asking it anyway is
            # the one bring-up that did not come back on the test box -- not attempted
            transport_used, transport_note = "host", (f"fallback: {N} ranks share {max(ngpu, 1)} GPU(s), RCCL needs one device per "
                                                      "rank -- not attempted; this line is a rehearsal of the N > 1 path, not a measurement")
            if rank == 0:
                print(f"[bench] {transport_note}", file=sys.stderr, flush=True)
        if transport_used == "rccl":
            # RCCL inside libptnn has never run on more than one GPU (no multi-GPU box was ever available to the build): every stage
            # is bounded (PTNN_COMM_TIMEOUT_S) and agreed on by all ranks, and a bring-up or first-run failure anywhere moves
            # EVERY rank to the host-staged transport -- the line then says so and is a degraded figure, not an RCCL measurement.
            why = None
            box = [None, None]
            if rank == 0:
                try:
                    box[0] = _lib.comm_unique_id()
                except Exception as e:                          # noqa: BLE001
                    box[1] = f"rank 0 ncclGetUniqueId: {e!r}"
            dist.broadcast_object_list(box, src=0)
            if box[0] is None:
                why = box[1]
            else:
                err = None
                try:
                    s.comm_init(box[0], rank, N)
                    s.comm_set_mode(mode)
                    lad.whole_run()                             # preflight: the first collectives of this communicator, untimed
                except Exception as e:                          # noqa: BLE001
                    err = f"rank {rank}: {e!r} (last stage: {_lib.comm_last_stage()})"
                if not all_ok(err is None):
                    errs = [None] * N
                    dist.all_gather_object(errs, err)
                    why = "; ".join(x for x in errs if x) or "a peer failed"
            if why is not None:
                if rank == 0:
                    print(f"[bench] RCCL path failed, every rank falls back to the host-staged transport: {why}", file=sys.stderr, flush=True)
                try:
                    s.close()
                except Exception:                               # noqa: BLE001
                    pass
                lad = Ladder(wl, a, train, test, rank, N, device, shared_device=shared_dev)
                s = lad.s
                transport_used, transport_note = "host", "fallback after an RCCL failure: " + why[:600]
                dm.arm_hard_exit()                             # an RCCL left half-initialised in this process may never let it exit
        if transport_used == "host":
            s.comm_init_host(rank, N, *dm.gloo_transport(dist))
            s.comm_set_mode(mode)

    def fence():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(W):
        lad.whole_run()
    s.kernel_time(reset=True)
    fence()
    t0 = time.perf_counter()
    for _ in range(K):
        lad.whole_run()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    launches, kms = s.kernel_time()
    comm = None
    if dist is not None:
        # what the communicator itself saw, from every rank (ptnn_comm_info / ptnn_comm_stats), not what argv asked for
        mine = dict(s.comm_info(), **s.comm_stats())
        allc = [None] * N
        dist.all_gather_object(allc, mine)
        comm = {"transport": sorted({c["transport"] for c in allc})[0] if len({c["transport"] for c in allc}) == 1 else [c["transport"] for c in allc],
                "nranks_seen": sorted({c["nranks"] for c in allc})[0] if len({c["nranks"] for c in allc}) == 1 else [c["nranks"] for c in allc],
                "device_ids": [c["device"] for c in allc], "distinct_devices": len({c["device"] for c in allc}),
                "exchange": allc[0]["mode"], "bytes_sent": [c["bytes_sent"] for c in allc], "bytes_received": [c["bytes_received"] for c in allc],
                "rounds": allc[0]["rounds"], "launcher": "self" if os.environ.get("PTNN_BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if launched_by_env else "none (one process)")}
    nsw, tot, rounds = s.swap_stats()            # counters restart with every run: these are the last run's
    st = s.state()
    info = s.describe()

    # ---- untimed extras (rank 0): throughput of the half of a run the reference keeps after burn-in; dependent-chain analysis
    extras = {}
    if rank == 0 and N == 1 and not a.no_extras:
        half = (S // 2 // si) * si + (1 if wl["task"] == 0 else 0)      # REG hands off after step k si, CLS after k si - 1
        tb = time.perf_counter()
        tm = lad.whole_run(split=half)
        te = time.perf_counter()
        extras["first_half"] = {"value": R * half / (tm - tb), "unit": "samples/s", "mh_steps": f"1..{half}"}
        extras["kept_half"] = {"value": R * (S - 1 - half) / (te - tm), "unit": "samples/s", "mh_steps": f"{half + 1}..{S - 1}",
                               "note": "the samples the reference keeps (burn_in 0.5, REG:949,1004)"}
        if info["slots_per_round"] > 1 and wl["lg"] and wl["topo"][1] <= 64 and S * R * 4 < (1 << 28):
            # what ONE sequential SGD epoch costs here and now: timed inside a kernel of this process (no stamp count from another
            # build, no assumed clock)
            epoch_ms = s.time_sgd_epoch(lad.w0[0], reps=200)
            extras["dependent_chain"] = dependent_chain(s.traces(pos_w=False)["accept"], si, info["slots_per_round"], epoch_ms)
        if info["schedule"] == "prefetching-tree":
            # what a round of the tree cannot do without, timed inside a kernel of this process: one forward pass of a node's proposal
            # by one work-group, and one record granule from one work-group to another (through the path the records take)
            fw_ms, hop_ms, local = s.time_tree_round(lad.w0[0], reps=200, xcd_local=True)
            depth = info["slots_per_round"]
            rounds = -(-si // depth)                            # a round commits `depth` steps whatever the decisions
            extras["tree_round"] = {"forward_pass_ms": fw_ms, "record_hop_ms": hop_ms, "records_through_xcd_l2": local, "steps_per_round": depth,
                                    "rounds_per_interval": rounds, "floor_ms_per_interval": rounds * (fw_ms + hop_ms),
                                    "source": "measured in this process: ptnn_time_tree_round, 200 forward passes of one work-group / 200 granule "
                                              "round trips between two work-groups of one XCD, in-kernel constant-rate counter",
                                    "note": "a round = every node's forward pass at the same time, then every work-group needs every record: "
                                            "no schedule of this tree is faster than rounds x (one forward pass + one record hop)"}
        if a.workload == "sunspot64" and not a.rw and not a.replicas:
            # the other noise mode on the same workload: a few whole runs, timed the same way
            other = 1 - a.shared_noise
            lad2 = Ladder(wl, a, train, test, rank, N, device, shared_noise=other)
            lad2.whole_run()
            t2 = time.perf_counter()
            for _ in range(5):
                lad2.whole_run()
            d2 = time.perf_counter() - t2
            n2, tot2, _ = lad2.s.swap_stats()
            extras["other_noise_mode"] = {"shared_noise": other, "value": R * (S - 1) * 5 / d2, "unit": "samples/s", "runs": 5,
                                          "swap_accept_pct": 100.0 * n2 / max(tot2, 1),
                                          "mh_accept_pct": float(100.0 * np.mean(lad2.s.state()["num_accepted"]) / max(S - 1, 1)),
                                          "note": "shared_noise 1 = the reference's behaviour (its forked chains inherit one RNG state, "
                                                  "REG:709-712) and the drop-in's default; 0 = independent Philox streams per chain"}
            lad2.s.close()
        if not a.rw and not a.replicas and R * S * (lad.P + 24) * 4 <= (1 << 30):
            # the whole drop-in call the reference's own figure is taken over (REG:1019-1022): run_chains() with the trace download
            # over PCIe, the per-chain result files and show_results -- never `value` (which has its inputs and outputs in HBM).
            # For the workloads whose traces fit the drop-in's pinned images (1 GiB): Sunspot, Iris, Mackey-Glass
            try:
                import shutil
                import tempfile
                tmp = tempfile.mkdtemp(prefix="ptnn_bench_")
                if wl["task"] == 0:
                    from ptnn_amd.pt_timeseries_regression import ParallelTempering
                    pt = ParallelTempering(bool(wl["lg"]), wl["lr"], train, test, list(wl["topo"]), R, wl["maxtemp"], R * S, si, 0.5, tmp, seed=SEED,
                                           shared_noise=bool(a.shared_noise))
                else:
                    from ptnn_amd.pt_classification import ParallelTempering
                    pt = ParallelTempering(bool(wl["lg"]), wl["lr"], train, test, list(wl["topo"]), R, wl["maxtemp"], R * S, si, tmp, seed=SEED,
                                           shared_noise=bool(a.shared_noise))
                for sub in ("predictions", "posterior", "posterior/pos_w", "posterior/pos_likelihood", "posterior/accept_list"):
                    pt.make_directory(os.path.join(tmp, sub))
                pt.initialize_chains(0.5)
                t3 = time.perf_counter()
                pt.run_chains()
                d3 = time.perf_counter() - t3
                nbytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tmp) for f in fs)
                shutil.rmtree(tmp, ignore_errors=True)
                tm = pt.timings
                extras["end_to_end"] = {"run_chains_s": d3, "value": R * (S - 1) / d3, "unit": "samples/s",
                                        "sampling_s": tm.get("sampling_s"), "trace_download_s": tm.get("fetch_s"),
                                        "files_and_results_s": tm.get("files_and_results_s"), "show_results_s": tm.get("show_results_s"),
                                        "files_queue_s": tm.get("chain_files_s"), "files_drain_s": tm.get("files_drain_s"),
                                        "result_file_bytes": nbytes, "overlapped": bool(tm.get("overlapped")),
                                        "launches_per_run": tm.get("launches_per_run", 1), "rows_landed_s": tm.get("rows_landed_s"), "segment_kernel_ms": tm.get("segment_kernel_ms"),
                                        "pcie_inclusive_samples_per_s": R * (S - 1) / max((tm.get("sampling_s") or 0) + (tm.get("fetch_s") or 0), 1e-9),
                                        "note": "the drop-in ParallelTempering(...).run_chains() of this workload, files included; reported, "
                                                "never `value`.  overlapped: the run is cut into launches_per_run launches and the trace "
                                                "rows of each go to the host and into the files while the next samples; sampling_s then "
                                                "ends when the last rows have landed (trace_download_s is inside it)"}
            except Exception as e:                              # noqa: BLE001  (a full /tmp must not cost the bench line)
                extras["end_to_end"] = {"error": repr(e)}

    if rank == 0:
        mh_steps = S - 1                                        # per replica and run
        value = R * N * mh_steps * K / dt
        avg_launch_s = (kms / max(launches, 1)) * 1e-3
        steps_per_launch = mh_steps * K / max(launches, 1)
        P = lad.P
        # mandatory HBM traffic of one MH step of one replica = the trace row the result files require,
        # 4 (P + 7) bytes (pos_w row + likeh + 2 rmse + 2 acc + accept count); a swap round adds 4 (P + 2) per replica
        rounds_per_launch = rounds * K / max(launches, 1)            # a persistent launch holds every swap round of its run
        bytes_per_launch = R * steps_per_launch * 4 * (P + 7) + rounds_per_launch * R * 4 * (P + 2)
        flops_per_launch = R * steps_per_launch * flops_per_step(wl["topo"], train.shape[0], test.shape[0], 0.5 if wl["lg"] else 0.0)
        traffic, traffic_tag = pmc_traffic(a.workload, info["kernel"], steps_per_launch) if (a.schedule, a.waves, a.groups, a.rw, a.bf16, a.replicas) == (0, 0, 0, False, False, 0) and N == 1 else (None, None)
        roof = {"bound": "hbm", "achieved": bytes_per_launch / avg_launch_s / 1e9 if launches else 0.0, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "traffic": traffic, "kernel": info["kernel"], "avg_launch_ms": avg_launch_s * 1e3,
                "launches": launches, "mh_steps_per_launch": steps_per_launch, "swap_rounds_per_launch": rounds_per_launch,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "traffic_source": f"profiles/current_pmc_{a.workload}.json ({traffic_tag}: rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per launch, "
                                  "separate passes of this command)" if traffic else None,
                "valu_tflops": flops_per_launch / avg_launch_s / 1e12 if launches else 0.0,
                "busy_cus": min(info["grid_blocks"], info["num_cus"]), "num_cus": info["num_cus"]}
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        roof["valu_frac"] = roof["valu_tflops"] / VALU_PEAK_TFLOPS
        roof["valu_frac_of_busy_cus"] = roof["valu_frac"] * info["num_cus"] / max(roof["busy_cus"], 1)
        roof["traffic_measured_in_run"] = False if traffic else None      # recorded by profiles/collect.sh (rocprofv3 PMC passes), not by this run
        dc = extras.get("dependent_chain")
        if dc and launches:
            intervals_per_launch = max(1.0, steps_per_launch / si)
            measured = avg_launch_s * 1e3 / intervals_per_launch
            roof["chain"] = {"floor_ms": dc["chain_floor_ms_per_interval"], "measured_ms": measured,
                             "frac": dc["chain_floor_ms_per_interval"] / measured if measured else None, "unit": "ms per swap interval",
                             "sgd_epoch_ms": dc["sgd_epoch_ms"],
                             "note": "floor = (accepted steps of the slowest replica of an interval, mean over the run's intervals) x one "
                                     "sequential SGD epoch timed in this process; measured = dominant kernel time per swap interval"}
        trd = extras.get("tree_round")
        if trd and launches:
            intervals_per_launch = max(1.0, steps_per_launch / si)
            measured = avg_launch_s * 1e3 / intervals_per_launch
            roof["tree"] = {"floor_ms": trd["floor_ms_per_interval"], "measured_ms": measured, "frac": trd["floor_ms_per_interval"] / measured if measured else None,
                            "unit": "ms per swap interval", "forward_pass_ms": trd["forward_pass_ms"], "record_hop_ms": trd["record_hop_ms"],
                            "rounds_per_interval": trd["rounds_per_interval"],
                            "note": "floor = rounds per interval x (one forward pass of a node by one work-group + one record granule between two "
                                    "work-groups), both timed in this process; measured = dominant kernel time per swap interval"}
        # ONE number per configuration: the fraction of the bound that actually binds this schedule
        if "chain" in roof:
            roof["binding"] = {"kind": "dependent chain of sequential SGD epochs", "frac": roof["chain"]["frac"], "see": "roofline.chain"}
        elif "tree" in roof:
            roof["binding"] = {"kind": "rounds of the prefetching tree (forward pass + record exchange)", "frac": roof["tree"]["frac"], "see": "roofline.tree"}
        roof["note"] = ("instruction-issue / dependent-chain bound by construction (sequential SGD rows, arithmetic intensity far "
                        "above the machine balance); the HBM fraction is reported because BASELINE.json asks for it")
        if info.get("forward_mfma"):
            # configs 4 and 5: the forward pass is a per-replica GEMM (Ntr+Nte) x I x H on the matrix cores -- its own roofline beside it.
            # forward_mfma 2 = fp32 operands split into three bf16 terms, six partial products on v_mfma_f32_32x32x16_bf16 (fp32
            # accuracy; the fp32 matrix instruction runs at VALU rate and blocks the VALU on gfx950): `achieved` stays the
            # ALGORITHMIC fp32 flops against the fp32 dense peak, `pipe` is what the bf16 pipe executes for them against ITS peak
            I_, H_ = wl["topo"][0], wl["topo"][1]
            gemm = 2.0 * (train.shape[0] + test.shape[0]) * I_ * H_
            split = info["forward_mfma"] == 2
            peak = MFMA_BF16_PEAK_TFLOPS if a.bf16 else VALU_PEAK_TFLOPS
            ach = R * steps_per_launch * gemm / avg_launch_s / 1e12 if launches else 0.0
            roof["mfma"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                            "instruction": "v_mfma_f32_32x32x16_bf16 on three-way split fp32 operands (6 partial products per k-step of 16)"
                                           if split else ("v_mfma_f32_32x32x16_bf16 (operands rounded)" if a.bf16 else "v_mfma_f32_32x32x2_f32"),
                            "note": "flops of the forward GEMM only, over the whole fused launch (which also runs the SGD epochs, the "
                                    "random tape and the vector streams)"}
            if split:
                kbf = 16 * ((I_ + 15) // 16 if I_ % 16 >= 7 else I_ // 16)      # SplitK: k extent on the bf16 instruction
                rows_pad = (train.shape[0] + test.shape[0] + 31) // 32 * 32
                pipe = 6.0 * 2.0 * rows_pad * kbf * ((H_ + 31) // 32 * 32)
                roof["mfma"]["pipe"] = {"achieved": R * steps_per_launch * pipe / avg_launch_s / 1e12 if launches else 0.0,
                                        "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s (bf16, padded tiles, 6 products)"}
                roof["mfma"]["pipe"]["frac"] = roof["mfma"]["pipe"]["achieved"] / MFMA_BF16_PEAK_TFLOPS
        if "binding" not in roof:
            # cooperative and wide schedules: every CU runs one chain's step; what binds is vector issue (sigmoid / W2 epilogue, SGD
            # rows), not the matrix pipe
            roof["binding"] = {"kind": "fp32 vector issue on the busy CUs", "frac": roof["valu_frac_of_busy_cus"], "see": "roofline.valu_frac_of_busy_cus"}
        out = {
            "metric": "MCMC samples/sec (all replicas) + swap-accept rate; " + ("Sunspot 64-replica FNN" if a.workload == "sunspot64" else wl["desc"]),
            "value": value, "unit": "samples/s", "n_gpus": N, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": a.scaling if N > 1 else "weak", "vs_baseline": None, "dtype": "bf16" if a.bf16 else "f32", "data": data_desc,
            "config": {"workload": wl["desc"] + f"; swap every {si} MH steps; 1 bench step = 1 WHOLE RUN from the chain start: "
                                                f"S = {S} samples per replica ({S - 1} MH steps, {S // si} swap rounds, chain start-up included)",
                       "replicas": R * N, "replicas_per_gpu": R, "samples_per_replica": S, "swap_interval": si,
                       "proposals": "langevin p=0.5" if wl["lg"] else "random-walk",
                       "noise": "shared tape (shared_noise=1: the reference's forked chains all inherit one RNG state, REG:709-712; the "
                                "drop-in's default)" if a.shared_noise else "independent Philox streams per chain (shared_noise=0)",
                       "schedule": info["schedule"],
                       "slots_per_round": info["slots_per_round"], "groups_per_replica": info["groups_per_replica"],
                       "block_threads": info["block_threads"], "lds_bytes": info["lds_bytes"], "exchange": info.get("exchange", "none"),
                       "transport": transport_used, **({"transport_note": transport_note} if transport_note else {})},
            "swap_accept_pct": 100.0 * nsw / max(tot, 1), "swap_rounds_per_run": rounds,
            "mh_accept_pct": float(100.0 * np.mean(st["num_accepted"]) / max(S - 1, 1)),
            "roofline": roof,
            "timed_region": "W warm-up runs, then K whole runs back to back: chain restart (ptnn_set_state), all MH steps, all swap rounds, "
                            "trace rows written to HBM; NOT in it: the device-to-host copy of the traces and the result files "
                            "(ParallelTempering.timings reports those; DESIGN.md 8)",
        }
        out.update(extras)
        if comm is not None:
            out["comm"] = comm
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["speedup_vs_cpu_baseline"] = value / cpu["value"]
            if isinstance(cpu.get("c_port"), dict) and cpu["c_port"].get("value"):
                out["speedup_vs_c_port"] = value / cpu["c_port"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        s.close()
        dist.destroy_process_group()
        if transport_note:
            # an RCCL whose bring-up was abandoned can leave threads behind that keep the interpreter from exiting (seen on the test
            # box: ncclGetUniqueId in a process that never completes an ncclCommInitRank); the line is out, leave without teardown
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
    else:
        s.close()


if __name__ == "__main__":
    main()
