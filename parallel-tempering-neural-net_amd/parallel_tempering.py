"""Host side of the drop-in: the public surface of the reference's `ParallelTempering` class.

Reference: REG = multicore-pt-regression/pt_timeseries_regression.py:487-875,
           CLS = multicore-pt-classification/pt_classification.py:497-897.

What `main()` touches (REG:995-1007, CLS:1080-1092) is kept: the constructor arguments, `make_directory`,
`initialize_chains(burn_in)`, `run_chains()` with its 11-tuple, the attributes `num_swap`,
`total_swap_proposals`, `temperatures`, `NumSamples`, `num_param`, and every file under `path`
(SURVEY.md section 8b).  What happens in between -- one forked process per replica, queues and events --
is replaced by libptnn.so: all replicas advance inside one HIP kernel -- one launch for the whole run with the
swap cascade inside it where every work-group is resident, else one launch per swap interval with the cascade
as a second kernel; this module only configures the run, fetches the traces and writes the files.
"""
import math
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _lib, ladder, philox

TASK_REG, TASK_CLS = _lib.TASK_REG, _lib.TASK_CLS


def _text_round(a, fmt, threads=8):
    """Values as np.loadtxt would read them back from np.savetxt(..., fmt=fmt) (show_results re-reads the files)."""
    return _lib.text_round(a, fmt, threads)


def stitch_by_temperature(tr, swap_log, handoff_steps, S):
    """Label swapping: the device records trace rows per chain slot and a slot's temperature changes at the swap rounds.  Row
    i + 1 is written by MH step i, so the rows up to the hand-off step + 1 belong to the assignment before the round; round k
    hands temperature t to the chain that held temperature swap_log[k][t].  -> (traces keyed by temperature, final holder[t])."""
    R = next(v for v in tr.values() if v is not None).shape[0]
    holder = np.arange(R)                                    # holder[t] = slot whose chain holds temperature t
    out = {k: (None if v is None else np.empty_like(v)) for k, v in tr.items()}
    row0 = 0
    for k, i_k in enumerate(list(handoff_steps) + [None]):
        row1 = S if i_k is None else i_k + 2
        for key, v in tr.items():
            if v is not None:
                out[key][:, row0:row1] = v[holder, row0:row1]
        if i_k is None:
            break
        holder = holder[np.asarray(swap_log[k])]
        row0 = row1
    return out, holder


def overlap_cuts(S, swap_interval, chunks):
    """MH-step counts at which an overlapped run_chains() ends its launches: at most `chunks` launches of whole swap intervals
    (near-equal), strictly increasing, the last one at S - 1 (all steps; ptnn_run then also runs the phantom round if one is due)."""
    si = max(1, int(swap_interval))
    n_int = max(1, (S - 1) // si)
    K = max(1, min(int(chunks), n_int))
    return sorted({min(S - 1, si * max(1, round(n_int * (c + 1) / K))) for c in range(K - 1)} | {S - 1})


class ParallelTemperingBase:
    task = None                       # set by the two drop-in subclasses
    rmse_fmt = None                   # REG '%1.8f' (REG:462-464), CLS '%1.2f' (CLS:473-475)

    def __init__(self, use_langevin_gradients, learn_rate, traindata, testdata, topology, num_chains, maxtemp,
                 NumSample, swap_interval, langevin_prob, path, *, seed=None, device=None, devices=None, exchange="auto",
                 transport=None, waves_per_replica=0, schedule=0, groups_per_replica=0, trace_capacity=0, swap_rule=0,
                 label_swap=False, shared_noise=True, write_files=True, io_threads=None, forward_bf16=0, overlap_chunks=8):
        # FNN chain variables (REG:491-494)
        self.traindata = traindata
        self.testdata = testdata
        self.topology = topology
        self.num_param = (topology[0] * topology[1]) + (topology[1] * topology[2]) + topology[1] + topology[2]
        # parallel tempering variables (REG:496-507)
        self.swap_interval = swap_interval
        self.path = path
        self.maxtemp = maxtemp
        self.langevin_prob = langevin_prob
        self.num_swap = 0
        self.total_swap_proposals = 0
        self.num_chains = num_chains
        self.chains = []
        self.temperatures = []
        self.NumSamples = int(NumSample / self.num_chains)
        self.learn_rate = learn_rate
        self.use_langevin_gradients = use_langevin_gradients
        # build-specific knobs (keyword only; defaults reproduce the reference's behaviour)
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")       # the reference never seeds its generators
        self.seed = int(seed)
        self.device = int(os.environ.get("PTNN_DEVICE", "0")) if device is None else int(device)
        # devices=[0, 1, ...]: the ladder is cut into len(devices) equal contiguous blocks, one per GPU; the swap rounds exchange
        # over RCCL inside libptnn (where the reference forks one process per chain and pipes every vector through the parent,
        # REG:694-771).  $PTNN_DEVICES="0,1,2,3" does the same for an unmodified driver script.  `exchange`: "auto", "gather"
        # or "boundary" (include/ptnn.h); `transport`: None = RCCL when the devices are distinct, host-staged otherwise;
        # "auto" = try RCCL whatever the list; both fall back to host-staged (with a warning) when the RCCL bring-up fails; "rccl" / "host" = that one or an error.
        if devices is None and os.environ.get("PTNN_DEVICES"):
            devices = [int(v) for v in os.environ["PTNN_DEVICES"].split(",")]
        self.devices = None if devices is None else [int(d) for d in devices]
        if self.devices is not None and self.num_chains % len(self.devices) != 0:
            raise ValueError(f"num_chains = {self.num_chains} cannot be cut into {len(self.devices)} equal blocks (one per device)")
        self.exchange = {"auto": _lib.XCHG_AUTO, "gather": _lib.XCHG_GATHER, "boundary": _lib.XCHG_BOUNDARY}[exchange]
        self.transport = transport
        self.waves_per_replica = int(waves_per_replica)
        self.schedule = int(schedule)            # 0 auto, 1 cooperative, 2 speculative, 3 packed, 4 prefetching tree (include/ptnn.h)
        self.groups_per_replica = int(groups_per_replica)
        self.swap_rule = int(swap_rule)          # 0 = the reference's cascade; 1 = even/odd Metropolis exchange (not in the reference)
        # forward pass on the matrix cores (nets of 24..64 or a multiple of 32 > 64 hidden units): 0 = fp32 accuracy on split bf16
        # operands (default), 2 = the exact fp32 instruction (bit-identical to the other schedules), 1 = operands rounded to bf16 (study)
        self.forward_bf16 = int(forward_bf16)
        # True (default): all chains read ONE noise tape -- what the reference's forked chains do, which all inherit the parent's
        # numpy / random state (REG:709-712); the only mode that meets every statistical parity bound against the
        # reference's own runs (tests: F9).  False: every (chain, step) has its own Philox counter, the statistically sounder
        # choice (within-slot posterior variance comes out 1.3 - 1.8 x the reference's on high-acceptance chains).
        self.shared_noise = bool(shared_noise)
        # True: swap rounds permute which chain holds which temperature instead of moving (w, eta) between the temperature slots
        # (zero payload between GPUs; not in the reference).  The files stay keyed by temperature: the rows a
        # temperature's files hold are those of the chain that held it at the time (_stitch_by_temperature).
        self.label_swap = bool(label_swap)
        self.trace_capacity = int(trace_capacity)   # rows per replica kept in HBM (0 = all); smaller = streamed to the host
        self.write_files = bool(write_files)
        self.io_threads = io_threads or min(16, os.cpu_count() or 1)
        # run_chains() cuts the run into this many launches and lets the trace rows of each leave for the host -- and into the
        # per-chain files -- while the next is being sampled (one GPU, every row resident); 0: download and write after the last step
        self.overlap_chunks = int(overlap_chunks)
        self._img = None
        self.timings = {}
        self._sampler = None
        self._w0 = None

    # ------------------------------------------------------------------ ladder (REG:529-636)
    def default_beta_ladder(self, ndim, ntemps, Tmax):
        return ladder.default_beta_ladder(ndim, ntemps=ntemps, Tmax=Tmax)

    def assign_temperatures(self):
        """T_i = 1 / beta_i of the geometric ladder (the only spacing main() can reach, REG:615-628)."""
        betas = self.default_beta_ladder(2, ntemps=self.num_chains, Tmax=self.maxtemp)
        self.temperatures.extend(np.inf if b == 0 else float(1.0 / b) for b in betas)

    # ------------------------------------------------------------------ initialize_chains (REG:639-650)
    def initialize_chains(self, burn_in):
        self.burn_in = burn_in
        self.assign_temperatures()
        # w0 per chain: the reference draws np.random.randn(num_param) in the parent (REG:649); here the
        # draws come from Philox stream 3 keyed by (seed, chain) so that a run is reproducible from `seed`
        self._w0 = np.stack([philox.initial_weights(self.seed, r, self.num_param) for r in range(self.num_chains)])
        self._configure()

    def set_initial_weights(self, w0):
        """Override the initial weights (tests / resuming from a known state); shape [num_chains, num_param]."""
        w0 = np.asarray(w0, dtype=np.float64)
        if w0.shape != (self.num_chains, self.num_param):
            raise ValueError("w0 must be [num_chains, num_param]")
        self._w0 = w0
        if self._sampler is not None:
            self._sampler.set_state(self._w0, self.temperatures)

    def _pt_switch_step(self):
        # `i == pt_samples` with pt_samples = samples * 0.6 compares an int with a float (REG:301,320): it fires
        # only when the product is integral in double arithmetic
        pt_samples = self.NumSamples * 0.6
        return int(pt_samples) if pt_samples == int(pt_samples) else -1

    def _configure(self):
        I, H, O = (int(v) for v in self.topology)
        S = self.NumSamples
        if self.swap_interval < 1:
            raise ZeroDivisionError("integer division or modulo by zero")      # `i % self.swap_interval` (REG:427)
        train = np.asarray(self.traindata, dtype=np.float64)
        test = np.asarray(self.testdata, dtype=np.float64)
        if train.ndim != 2 or train.shape[1] <= I:
            raise IndexError(f"index {I} is out of bounds for axis 1 with size {train.shape[1] if train.ndim == 2 else 0}")
        lib = _lib.load_library()
        if not lib.ptnn_supports(self.task, I, H, O):
            raise _lib.PtnnError(f"no gfx950 kernel for task={self.task} topology={[I, H, O]} in {_lib.library_path()}: "
                                 f"add X({self.task}, {I}, {O}) to PTNN_SHAPES (csrc/ptnn_shapes.hpp) and to SHAPES in "
                                 f"__graft_entry__.py, then rebuild; n_hidden may be anything up to 512")
        config = dict(
            task=self.task, n_in=I, n_hidden=H, n_out=O, n_replicas_global=self.num_chains,
            n_samples=S, swap_interval=int(self.swap_interval), pt_switch_step=self._pt_switch_step(),
            use_langevin=1 if self.use_langevin_gradients is True else 0, waves_per_replica=self.waves_per_replica,
            schedule=self.schedule, groups_per_replica=self.groups_per_replica, trace_capacity=self.trace_capacity,
            swap_rule=self.swap_rule, shared_noise=int(self.shared_noise), label_swap=int(self.label_swap),
            forward_bf16=self.forward_bf16, l_prob=float(self.langevin_prob), learn_rate=float(self.learn_rate), step_w=0.025, step_eta=0.2,
            sigma_squared=25.0, nu_1=0.0, nu_2=0.0, seed=self.seed)
        if self.devices is not None and len(self.devices) > 1:
            from . import distributed
            self._sampler = distributed.LadderGroup(self.devices, exchange=self.exchange, transport=self.transport, **config)
        else:
            dev = self.device if self.devices is None else self.devices[0]
            self._sampler = _lib.Sampler(device_id=dev, n_replicas_local=self.num_chains, first_global_replica=0, **config)
        self._sampler.set_data(train, test)
        self._sampler.set_state(self._w0, self.temperatures)
        if self.swap_rule == 1 or self.label_swap:
            self._sampler.set_ladder(self.temperatures)
        self._img = None
        # (the images are pinned host memory the size of the device's trace arrays: taken up to 1 GiB, above that the resident path)
        image_bytes = self.num_chains * S * (self.num_param + 24) * 4
        if (self.overlap_chunks > 1 and isinstance(self._sampler, _lib.Sampler) and not self.label_swap and not (0 < self.trace_capacity < S)
                and image_bytes <= int(os.environ.get("PTNN_TRACE_IMAGE_MAX_BYTES", 1 << 30))):
            try:
                self._img = self._sampler.trace_image()      # pinned host images of the trace arrays, allocated here once
            except _lib.PtnnError:
                self._img = None                             # compact traces (wide nets): the resident path

    # ------------------------------------------------------------------ run_chains (REG:694-771)
    def run_chains(self, *, checkpoint_path=None, checkpoint_every=None, resume_from=None, max_steps=None):
        """The reference's run_chains().  Keyword-only extras (the reference has none): `checkpoint_path` +
        `checkpoint_every` (MH steps) write a resumable .npz (device state of the chains + the trace rows fetched so far)
        as the run proceeds; `resume_from` continues such a file bit for bit; `max_steps` stops after that many MH steps of
        this call (writing a checkpoint when a path is given) and returns None instead of the result tuple."""
        if self._sampler is None:
            raise RuntimeError("call initialize_chains(burn_in) before run_chains()")
        S = self.NumSamples
        open(self.path + '/num_exchange.txt', 'a').close()                      # REG:704: opened, never written
        t0 = time.perf_counter()
        cap = self.trace_capacity
        chunked = bool(cap and cap < S) or checkpoint_every or resume_from or max_steps
        if chunked:
            # the device keeps a ring of `cap` trace rows per replica (all S rows when cap == 0): drain it in windows
            parts, row, t_fetch = [], 0, 0.0
            if resume_from is not None:
                with np.load(resume_from) as z:
                    self._sampler.restore(z["blob"].tobytes())
                    parts.append({k[3:]: z[k] for k in z.files if k.startswith("tr_")})
                row = self._sampler.steps_done() + 1
                if parts[0]["accept"].shape[1] != row:
                    raise ValueError("checkpoint file is inconsistent: trace rows do not end at the saved step")
            window = (cap - 1) if (cap and cap < S) else S
            if checkpoint_every:
                window = min(window, int(checkpoint_every))
            budget = None if max_steps is None else int(max_steps)

            def save():
                tf = {k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0]}
                tmp = checkpoint_path + ".tmp.npz"
                np.savez(tmp, blob=np.frombuffer(self._sampler.checkpoint(), np.uint8), **{"tr_" + k: v for k, v in tf.items()})
                os.replace(tmp, checkpoint_path)
            while self._sampler.steps_done() < S - 1 and (budget is None or budget > 0):
                n = min(window, S - 1 - self._sampler.steps_done())
                if budget is not None:
                    n = min(n, budget)
                    budget -= n
                self._sampler.run(n)
                self._sampler.sync()
                tf = time.perf_counter()
                hi = self._sampler.steps_done() + 1
                parts.append(self._sampler.traces(row, hi - row))
                row = hi
                if checkpoint_path is not None and (checkpoint_every or budget == 0):
                    save()
                t_fetch += time.perf_counter() - tf
            if self._sampler.steps_done() < S - 1:
                return None                                                     # max_steps reached: resume later
            self._sampler.run(-1)                                               # phantom round, if due
            self._sampler.sync()
            tr = {k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0]}
            t1 = time.perf_counter() - t_fetch
            t2 = t1 + t_fetch
            self.num_swap, self.total_swap_proposals, self.rounds = self._sampler.swap_stats()
        elif self._img is not None:
            return self._run_overlapped(t0)
        else:
            self._sampler.run(-1)
            self._sampler.sync()
            t1 = time.perf_counter()
            self.num_swap, self.total_swap_proposals, self.rounds = self._sampler.swap_stats()
            tr = self._sampler.traces()
            t2 = time.perf_counter()
        if self.label_swap:
            tr = self._stitch_by_temperature(tr)
        # the per-chain files (REG:454-481) and what show_results derives from them (REG:775-871) are independent of each other
        # once the traces are on the host: one pool formats the 8 R + 3 files while this thread builds the return values
        with ThreadPoolExecutor(max_workers=self.io_threads) as ex:
            pending = self._write_chain_files(tr, ex) if self.write_files else []
            t3 = time.perf_counter()
            out = self.show_results(tr, _pool=ex, _pending=pending)
            t4 = time.perf_counter()
            for f in pending:
                f.result()                                   # an I/O error of any file surfaces here
        t5 = time.perf_counter()
        return self._finish_run(out, dict(sampling_s=t1 - t0, fetch_s=t2 - t1, chain_files_s=t3 - t2, show_results_s=t4 - t3, files_drain_s=t5 - t4,
                                          files_and_results_s=t5 - t2, overlapped=False))

    def _finish_run(self, out, timings):
        nlaunch, kms = self._sampler.kernel_time()
        self.timings = dict(timings, segment_launches=nlaunch, segment_kernel_ms=kms,
                            samples_per_s=self.num_chains * (self.NumSamples - 1) / max(timings["sampling_s"], 1e-12))
        pos_w, fx_train, fx_test, rmse_train, rmse_test, acc_train, acc_test, likelihood_vec, accept_vec, accept = out
        swap_perc = self.num_swap * 100 / self.total_swap_proposals            # ZeroDivisionError when no round ran (REG:769)
        return (pos_w, fx_train, fx_test, rmse_train, rmse_test, acc_train, acc_test, likelihood_vec, swap_perc,
                accept_vec, accept)

    # ------------------------------------------------------------------ run_chains with the download and the files behind the sampling
    def _run_overlapped(self, t0):
        """The run in `overlap_chunks` launches (whole swap intervals each; the chain does not depend on the cut: ptnn_run); the
        trace rows of a launch are copied into the pinned images by a second stream as soon as it ends, and formatted into the
        per-chain files (append mode) by the pool, while the next launch samples.  Same bytes in every file as the resident path."""
        s, S, si = self._sampler, self.NumSamples, max(1, int(self.swap_interval))
        pos_img, rows_img = self._img
        ends = overlap_cuts(S, si, self.overlap_chunks)
        tickets, done, row = [], 0, 0
        for b in ends:
            s.run(-1 if b == S - 1 else b - done)           # queued, not waited for
            done = b
            tickets.append((s.trace_fetch(row, b + 1 - row), row, b + 1))       # rows 0 .. b exist once step b - 1 has run
            row = b + 1
        # one writer thread takes the windows in order (a file's pieces must follow each other); each window is one call into the C
        # library, which spreads its 7 R files over the I/O threads
        with ThreadPoolExecutor(max_workers=self.io_threads) as ex, ThreadPoolExecutor(max_workers=1) as writer:
            pending, landed = [], []
            for tk, lo, hi in tickets:
                s.trace_wait(tk)
                landed.append(round(time.perf_counter() - t0, 6))
                if self.write_files:
                    pending.append(writer.submit(self._write_chain_rows, pos_img, rows_img, lo, hi))
            s.sync()                                         # a failed run surfaces here
            t1 = time.perf_counter()
            self.num_swap, self.total_swap_proposals, self.rounds = s.swap_stats()
            zeros = np.zeros((self.num_chains, S), np.float32)
            tr = {"pos_w": pos_img, "likeh": rows_img[:, :, 0], "rmse_train": rows_img[:, :, 1], "rmse_test": rows_img[:, :, 2],
                  "acc_train": zeros if self.task == TASK_REG else rows_img[:, :, 3],      # REG:403 (the slot carries eta, ptnn.h)
                  "acc_test": rows_img[:, :, 4], "accept": rows_img[:, :, 5].view(np.int32)}
            if self.write_files:
                self._final_accepted = s.state()["num_accepted"]
                for r, T in enumerate(self.temperatures):
                    acc_ratio = int(self._final_accepted[r]) / (S * 1.0) * 100     # REG:447
                    pending.append(ex.submit(_lib.savetxt, f'{self.path}/posterior/accept_list/chain_{T}_accept.txt', np.array([acc_ratio]), '%1.4f'))
            t3 = time.perf_counter()
            out = self.show_results(tr, _pool=ex, _pending=pending)
            t4 = time.perf_counter()
            for f in pending:
                f.result()                                   # an I/O error of any file surfaces here
        t5 = time.perf_counter()
        return self._finish_run(out, dict(sampling_s=t1 - t0, fetch_s=0.0, chain_files_s=t3 - t1, show_results_s=t4 - t3, files_drain_s=t5 - t4,
                                          files_and_results_s=t5 - t1, overlapped=True, launches_per_run=len(ends), rows_landed_s=landed))

    def _write_chain_rows(self, pos_img, rows_img, lo, hi):
        """Rows [lo, hi) of every per-chain trace file (REG:454-481): one call into the C library, which spreads the 7 R files over
        the I/O threads (pieces of a file are written in order: the caller runs these calls one after the other)."""
        n, R = hi - lo, self.num_chains
        rows = rows_img[:, lo:hi]                                                 # [R, n, 8] view of the image
        likeh = np.zeros((R, n, 2), dtype=np.float32)
        likeh[:, :, 0] = rows[:, :, 0]
        if lo == 0:
            likeh[:, 0, 1] = -100.0                                               # row 0 = [-100, -100] (REG:293)
        # accept_list[i+1] holds the count BEFORE step i (REG:380): small integers, exact in float32
        accept = rows[:, :, 5].view(np.int32).astype(np.float32)
        acc_tr = np.zeros((R, n, 1), np.float32) if self.task == TASK_REG else rows[:, :, 3:4]
        big, small = [], []
        for r, T in enumerate(self.temperatures):
            tn = str(T)
            big.append((f'{self.path}/posterior/pos_w/chain_{tn}.txt', pos_img[r, lo:hi], '%.18e'))
            small += [
                (f'{self.path}/predictions/rmse_test_chain_{tn}.txt', rows[r, :, 2:3], self.rmse_fmt),
                (f'{self.path}/predictions/rmse_train_chain_{tn}.txt', rows[r, :, 1:2], self.rmse_fmt),
                (f'{self.path}/predictions/acc_test_chain_{tn}.txt', rows[r, :, 4:5], '%1.2f'),
                (f'{self.path}/predictions/acc_train_chain_{tn}.txt', acc_tr[r], '%1.2f'),
                (f'{self.path}/posterior/pos_likelihood/chain_{tn}.txt', likeh[r], '%1.4f'),
                (f'{self.path}/posterior/accept_list/chain_{tn}.txt', accept[r], '%1.4f'),
            ]
        _lib.savetxt_batch(big + small, append=lo > 0, threads=self.io_threads)   # the big files first

    # ------------------------------------------------------------------ label swapping: rows per chain slot -> rows per temperature
    def _handoff_steps(self):
        """MH steps after which a swap round moved something: REG i % si == 0, i != 0 (REG:427); CLS (i + 1) % si == 0 (CLS:438)."""
        si, S = int(self.swap_interval), self.NumSamples
        if self.task == TASK_REG:
            return [i for i in range(1, S - 1) if i % si == 0]
        return [i for i in range(S - 1) if (i + 1) % si == 0]

    def _stitch_by_temperature(self, tr):
        out, self._final_holder = stitch_by_temperature(tr, self._sampler.swap_log(), self._handoff_steps(), self.NumSamples)
        return out

    # ------------------------------------------------------------------ per-chain files (REG:454-481)
    def _chain_file_jobs(self, tr):
        S = self.NumSamples
        jobs = []
        for r, T in enumerate(self.temperatures):
            tn = str(T)
            likeh = np.zeros((S, 2), dtype=np.float32)
            likeh[:, 0] = tr["likeh"][r]
            likeh[0, 1] = -100.0                                                 # row 0 = [-100, -100] (REG:293)
            acc_ratio = int(self._final_accepted[r]) / (S * 1.0) * 100                 # REG:447
            # float32 arrays go to the C writer as they are (every value printed as the double it converts to, like np.savetxt);
            # the biggest file first so that the pool's last job is a small one
            jobs += [
                (f'{self.path}/posterior/pos_w/chain_{tn}.txt', tr["pos_w"][r], '%.18e'),
                (f'{self.path}/predictions/rmse_test_chain_{tn}.txt', tr["rmse_test"][r], self.rmse_fmt),
                (f'{self.path}/predictions/rmse_train_chain_{tn}.txt', tr["rmse_train"][r], self.rmse_fmt),
                (f'{self.path}/predictions/acc_test_chain_{tn}.txt', tr["acc_test"][r], '%1.2f'),
                (f'{self.path}/predictions/acc_train_chain_{tn}.txt', tr["acc_train"][r], '%1.2f'),
                (f'{self.path}/posterior/pos_likelihood/chain_{tn}.txt', likeh, '%1.4f'),
                (f'{self.path}/posterior/accept_list/chain_{tn}_accept.txt', np.array([acc_ratio]), '%1.4f'),
                (f'{self.path}/posterior/accept_list/chain_{tn}.txt', tr["accept"][r], '%1.4f'),
            ]
        return jobs

    def _write_chain_files(self, tr, pool):
        """Queues every per-chain file on `pool`; returns the futures."""
        # accept_list[i+1] holds the count BEFORE step i (REG:380); the percentage file uses the final count
        self._final_accepted = self._sampler.state()["num_accepted"]
        if self.label_swap:                                  # per temperature: the count of the chain that holds it at the end
            self._final_accepted = self._final_accepted[self._final_holder]
        jobs = self._chain_file_jobs(tr)
        jobs.sort(key=lambda j: -np.asarray(j[1]).size)
        return [pool.submit(_lib.savetxt, *j) for j in jobs]

    # ------------------------------------------------------------------ show_results (REG:775-871 / CLS:780-893)
    def _likelihood_rows(self, burnin):
        raise NotImplementedError

    def show_results(self, tr=None, _pool=None, _pending=None):
        if tr is None:
            tr = self._sampler.traces()
            if self.label_swap:
                tr = self._stitch_by_temperature(tr)
        S, R = self.NumSamples, self.num_chains
        burnin = int(S * self.burn_in)
        th = self.io_threads
        # the reference re-reads the per-chain text files, so every value below has been through their format
        # ('%.18e' round-trips a float32 exactly: the posterior matrix is the traces themselves, cut, widened and transposed)
        posterior = _lib.posterior_matrix(tr["pos_w"], burnin, th)              # (P, R (S - burnin)) float64
        rmse_train = _text_round(tr["rmse_train"][:, burnin:], self.rmse_fmt, th)
        rmse_test = _text_round(tr["rmse_test"][:, burnin:], self.rmse_fmt, th)
        acc_train = _text_round(tr["acc_train"][:, burnin:], '%1.2f', th)
        acc_test = _text_round(tr["acc_test"][:, burnin:], '%1.2f', th)
        accept_list = tr["accept"].astype(np.float64)
        lo = self._likelihood_rows(burnin)
        likelihood_vec = np.zeros((R * (S - lo), 2))                             # rows: chain after chain
        likelihood_vec[:, 0] = _text_round(tr["likeh"][:, lo:], '%1.4f', th).reshape(-1)
        if lo == 0:
            likelihood_vec[::S, 1] = -100.0
        accept_percent = np.zeros((R, 1))                                        # never filled (REG:780,860)

        fx_train_all = np.zeros((R, S - burnin, np.asarray(self.traindata).shape[0]))
        fx_test_all = np.zeros((R, S - burnin, np.asarray(self.testdata).shape[0]))
        rmse_train = rmse_train.reshape(R * (S - burnin), 1)
        acc_train = acc_train.reshape(R * (S - burnin), 1)
        rmse_test = rmse_test.reshape(R * (S - burnin), 1)
        acc_test = acc_test.reshape(R * (S - burnin), 1)
        accept_vec = accept_list
        accept = np.sum(accept_percent) / R
        if self.write_files:
            jobs = [(self.path + '/likelihood.txt', likelihood_vec, '%1.5f'), (self.path + '/accept_list.txt', accept_list, '%1.2f'),
                    (self.path + '/acceptpercent.txt', np.array([accept]), '%1.2f')]
            if _pool is not None:
                _pending.extend(_pool.submit(_lib.savetxt, *j) for j in jobs)
            else:
                for j in jobs:
                    _lib.savetxt(*j)
        return (posterior, fx_train_all, fx_test_all, rmse_train, rmse_test, acc_train, acc_test, likelihood_vec,
                accept_vec, accept)

    def make_directory(self, directory):
        if not os.path.exists(directory):
            os.makedirs(directory)
